// The direct 3x3 kernels of dconv3_b3.hip on TWO fp16 planes and THREE products (the arithmetic of igemm_f16x2.hip): the same source,
// compiled with DC_H2 -- see the head of dconv3_b3.hip.  Entry points catseg_dconv3_f16x2_wimg_bytes / _prep_batch,
// catseg_dconv3_f16x2, catseg_dconv3_bnbwd_f16x2.
#define DC_H2 1
#include "dconv3_b3.hip"
