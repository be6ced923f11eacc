// The fused matrix-core forward at d = 128 (four 32-column blocks): a second translation unit of
// gdn_forward_dense.hip so that the two families compile in parallel.  Only the fused kernel, its plan
// kernel and their dispatcher are instantiated here (entry: gdn_dense_fused_op_d128).
#define GDN_DENSE_EXTRA_DC 4
#include "gdn_forward_dense.hip"
