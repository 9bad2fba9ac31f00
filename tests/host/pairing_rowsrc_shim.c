/* test shim: the host-side tables of the final-pairing witness (sipp_amd/csrc/pairing_rowsrc.h) as a plain C library */
#include "pairing_rowsrc.h"

int shim_rows(void) { return AIR_PAIRING_ROWS; }
int shim_elems(void) { return PE_N; }
int shim_pool(void) { return PP_N; }
int shim_pool_row(void) { return PP_ROW; }
int shim_pool_gc(void) { return PP_GC; }
int shim_sources(uint16_t *src) { return pairing_row_sources(src); }
int shim_log_rows(int16_t *oprow, int16_t *steprow, int max_steps) { return pairing_log_rows(oprow, steprow, max_steps); }
int shim_elem_col(int u8, int e) { return pairing_elem_col(u8 ? AIR_PAIRING_LAYOUT_U8 : AIR_PAIRING_LAYOUT_U16, e); }
