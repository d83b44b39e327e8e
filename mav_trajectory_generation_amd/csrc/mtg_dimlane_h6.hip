// Dimension-in-lane instantiations for the remaining chain lengths of the N = 12 standard shape (mtg_dimlane_more_h6.inc).
// Compiled like mtg_dimlane.hip; registered with mtg_find_dimlane through mtg_dimlane_more_h6().
#define MTG_DL_SINGLE_POLICY 1
#define MTG_DL_TABLE_FN mtg_dimlane_more_h6
#define MTG_DL_TABLE_INC "mtg_dimlane_more_h6.inc"
#include "mtg_dimlane_table.h"
