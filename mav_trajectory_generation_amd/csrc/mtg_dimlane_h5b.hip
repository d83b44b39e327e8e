// Dimension-in-lane instantiations for K = 17 .. 31 of the N = 10 standard shape (mtg_dimlane_more_h5b.inc).
// Compiled like mtg_dimlane.hip; registered with mtg_find_dimlane through mtg_dimlane_more_h5b().
#define MTG_DL_SINGLE_POLICY 1
#define MTG_DL_TABLE_FN mtg_dimlane_more_h5b
#define MTG_DL_TABLE_INC "mtg_dimlane_more_h5b.inc"
#include "mtg_dimlane_table.h"
