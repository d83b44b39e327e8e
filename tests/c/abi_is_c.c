/* The drop-in boundary must be a C ABI: this file is compiled as C99 (gcc -std=c99 -pedantic) against
 * include/mtg_hip.h and linked with libmtg_hip.so.  It runs without a GPU: only argument validation paths. */
#include <stdio.h>
#include <string.h>

#include "mtg_hip.h"

int main(void) {
  mtg_context* ctx = NULL;
  mtg_plan_info info;
  int rc;
  memset(&info, 0, sizeof(info));
  if (strcmp(mtg_status_string(MTG_OK), "ok") != 0) return 1;
  if (mtg_context_create(0, NULL, NULL) != MTG_ERR_INVALID_ARGUMENT) return 2;
  rc = mtg_context_create(0, NULL, &ctx);
  if (rc != MTG_OK && rc != MTG_ERR_NO_DEVICE) return 3; /* no silent CPU fallback: either a device or an error */
  if (rc == MTG_ERR_NO_DEVICE && ctx != NULL) return 4;
  if (mtg_plan_get_info(NULL, &info) != MTG_ERR_INVALID_ARGUMENT) return 5;
  if (mtg_solve_linear(NULL, 1, NULL, NULL, NULL, NULL, NULL, NULL, 0) != MTG_ERR_INVALID_ARGUMENT) return 6;
  if (ctx) mtg_context_destroy(ctx);
  printf("C ABI ok (context_create rc=%d)\n", rc);
  return 0;
}
