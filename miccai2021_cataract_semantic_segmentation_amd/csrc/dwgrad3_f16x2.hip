// The direct backward-weight kernel of dwgrad3_b3.hip on TWO fp16 planes and THREE products: the same source compiled with DW_H2 (see
// the head of dwgrad3_b3.hip).  Entry point catseg_dwgrad3_f16x2.
#define DW_H2 1
#include "dwgrad3_b3.hip"
