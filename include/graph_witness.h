/* C-ABI of the calc-witness path -- drop-in for the symbol the reference exports.
 *
 * Replaces: reference include/graph_witness.h:7-28 (types, gw_calc_witness prototype, gw_free_status)
 * and its implementation src/lib.rs:28-111.  Same symbol name, argument order, ownership rules and
 * error messages; a caller compiled against the reference's own header links against
 * libcircom_witnesscalc_amd.so unchanged.
 *
 * Differences from the reference, by design:
 *   - evaluation runs on an MI355X (HIP); there is no CPU fallback -- without a usable HIP device the call
 *     fails with status ERROR ("no HIP device ...");
 *   - situations in which the reference panics across the FFI boundary (malformed graph, unknown input
 *     key, input length mismatch, invalid JSON, Shl overflow, bit-op result == r) return 1 with a message;
 *   - on success status is {OK, NULL} and nothing is printed (the reference leaves {ERROR,"test error"}
 *     and prints "OK", src/lib.rs:106-108); export GW_REFERENCE_QUIRKS=1 to reproduce both.
 */
#ifndef CWC_AMD_GRAPH_WITNESS_H
#define CWC_AMD_GRAPH_WITNESS_H

#include <stddef.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { OK = 0, ERROR = 1 } GW_ERROR_CODE;

typedef struct {
  GW_ERROR_CODE code;
  char *error_msg; /* malloc'ed, NUL-terminated, or NULL; always written by gw_calc_witness */
} gw_status_t;

/* inputs: NUL-terminated UTF-8 JSON object (reference src/lib.rs:195-247 value forms).
 * graph_data/graph_data_len: a whole `wtns.graph.001` file (reference src/storage.rs:214-249).
 * On success returns 0 and *wtns_data is a malloc'ed `.wtns` image of *wtns_len bytes (caller frees).
 * On failure returns 1, *wtns_data / *wtns_len untouched, status->error_msg set (caller frees). */
int gw_calc_witness(const char *inputs, const void *graph_data, const size_t graph_data_len,
                    void **wtns_data, size_t *wtns_len, const gw_status_t *status);

#ifndef GW_NO_INLINE_FREE_STATUS
/* reference include/graph_witness.h:23-28 defines this in the header; kept header-only here too */
static inline void gw_free_status(gw_status_t *status) {
  if (status->error_msg != NULL) free(status->error_msg);
}
#endif

#ifdef __cplusplus
}
#endif
#endif /* CWC_AMD_GRAPH_WITNESS_H */
