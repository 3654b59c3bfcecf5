/* A C host doing what the north star asks of the Rust host: it owns an RCCL communicator, loads the graph on the root
 * rank and lets the library broadcast the compiled program (gwb_graph_broadcast), then evaluates a batch on the handle it
 * got back.  One rank here (the GPU boxes of the test pool have one GPU); the non-root path = the same broadcast followed
 * by gwb_graph_import, which the import tests cover.  Usage: bcast_client <graph.bin> <inputs.json> */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "graph_witness_batch.h"

typedef struct { char internal[128]; } ncclUniqueId;
typedef int (*get_id_fn)(ncclUniqueId*);
typedef int (*init_rank_fn)(void**, int, ncclUniqueId, int);
typedef int (*destroy_fn)(void*);

static void* read_file(const char* path, size_t* len) {
    FILE* f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    *len = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    char* p = malloc(*len + 1);
    if (fread(p, 1, *len, f) != *len) { fclose(f); free(p); return NULL; }
    p[*len] = 0;
    fclose(f);
    return p;
}

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s <graph.bin> <inputs.json>\n", argv[0]); return 2; }
    void* rccl = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!rccl) rccl = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!rccl) { fprintf(stderr, "no RCCL: %s\n", dlerror()); return 3; }
    get_id_fn get_id = (get_id_fn)dlsym(rccl, "ncclGetUniqueId");
    init_rank_fn init_rank = (init_rank_fn)dlsym(rccl, "ncclCommInitRank");
    destroy_fn destroy = (destroy_fn)dlsym(rccl, "ncclCommDestroy");
    ncclUniqueId id;
    void* comm = NULL;
    if (!get_id || !init_rank || get_id(&id) != 0 || init_rank(&comm, 1, id, 0) != 0) { fprintf(stderr, "RCCL communicator init failed\n"); return 4; }
    /* the same through the library's own helpers (a host without a RCCL binding): id bytes, join, count, destroy */
    {
        gw_status_t st0;
        unsigned char idb[GWB_RCCL_UNIQUE_ID_BYTES];
        void* comm2 = NULL;
        if (gwb_rccl_unique_id(idb, &st0) != 0) { fprintf(stderr, "gwb_rccl_unique_id: %s\n", st0.error_msg); return 12; }
        if (gwb_rccl_comm_init(idb, 1, 0, &comm2, &st0) != 0) { fprintf(stderr, "gwb_rccl_comm_init: %s\n", st0.error_msg); return 13; }
        if (gwb_rccl_comm_ranks(comm2) != 1) { fprintf(stderr, "gwb_rccl_comm_ranks\n"); return 14; }
        if (gwb_rccl_comm_init(idb, 2, 5, &comm2, &st0) == 0) { fprintf(stderr, "a rank beyond the communicator accepted\n"); return 15; }
        gwb_free_status(&st0);
        if (destroy) destroy(comm);   /* the broadcast below runs on the library-made communicator */
        comm = comm2;
        destroy = NULL;
    }
    size_t glen, jlen;
    void* gdata = read_file(argv[1], &glen);
    char* json = read_file(argv[2], &jlen);
    if (!gdata || !json) { fprintf(stderr, "cannot read inputs\n"); return 5; }
    gw_status_t st;
    gwb_graph_t *g = NULL, *rep = NULL;
    if (gwb_graph_load(gdata, glen, &g, &st) != 0) { fprintf(stderr, "load: %s\n", st.error_msg); return 6; }
    if (gwb_graph_broadcast(g, 0, 1, 0, 0, comm, NULL, &rep, &st) != 0) { fprintf(stderr, "broadcast: %s\n", st.error_msg); return 7; }
    if (rep != g) { fprintf(stderr, "the root must get its own handle back\n"); return 8; }
    /* a misuse must fail cleanly */
    gwb_graph_t* bad = NULL;
    if (gwb_graph_broadcast(NULL, 0, 1, 0, 0, comm, NULL, &bad, &st) == 0) { fprintf(stderr, "root without a graph accepted\n"); return 9; }
    gwb_free_status(&st);
    gwb_graph_info_t info;
    gwb_graph_info(rep, &info);
    unsigned char* row = malloc(info.n_inputs * 32);
    unsigned char* wit = malloc(info.n_witness * 32);
    uint32_t set_status = 0;
    if (gwb_inputs_from_json(rep, json, row, &st) != 0) { fprintf(stderr, "inputs: %s\n", st.error_msg); return 10; }
    if (gwb_calc_witness_batch_host(rep, row, 1, wit, &set_status, &st) != 0 || set_status) { fprintf(stderr, "evaluate: %s\n", st.error_msg ? st.error_msg : "set status"); return 11; }
    for (size_t i = 0; i < info.n_witness; ++i) {
        for (int k = 31; k >= 0; --k) printf("%02x", wit[i * 32 + k]);
        printf("\n");
    }
    gwb_graph_free(g);
    if (destroy) destroy(comm);
    else gwb_rccl_comm_destroy(comm);
    return 0;
}
