/* Conformance client of the C-ABI boundary, in the shape of the reference's own caller
 * (examples/calc_witness.c:90-122: uninitialised gw_status_t, gw_free_status after success, caller frees the
 * witness).  usage: capi_client <inputs.json> <graph.bin> <witness.wtns>   (argument order of the C example) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "graph_witness.h"

static void *slurp(const char *path, size_t *len, int text) {
  FILE *f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  char *buf = malloc((size_t)n + 1);
  if (fread(buf, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read %s\n", path); exit(2); }
  fclose(f);
  buf[n] = 0;
  *len = (size_t)n;
  (void)text;
  return buf;
}

int main(int argc, char **argv) {
  if (argc != 4) { fprintf(stderr, "Usage: %s <inputs> <circuit_graph> <witness>\n", argv[0]); return 1; }
  size_t jl, gl;
  char *json = slurp(argv[1], &jl, 1);
  void *graph = slurp(argv[2], &gl, 0);
  void *wtns = NULL;
  size_t wtns_len = 0;
  gw_status_t status; /* deliberately uninitialised, like the reference example */
  int r = gw_calc_witness(json, graph, gl, &wtns, &wtns_len, &status);
  if (r != 0) {
    fprintf(stderr, "Error code: %i\n", status.code);
    if (status.error_msg != NULL) { printf("Error msg: %s\n", status.error_msg); free(status.error_msg); }
    return 1;
  }
  gw_free_status(&status);
  FILE *o = fopen(argv[3], "wb");
  if (!o || fwrite(wtns, 1, wtns_len, o) != wtns_len) { perror(argv[3]); return 2; }
  fclose(o);
  free(wtns);
  free(json);
  free(graph);
  return 0;
}
