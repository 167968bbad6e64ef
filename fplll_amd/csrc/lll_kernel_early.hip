// lll_kernel_early.hip — the LLL_EARLY_RED instantiations of lll_kernel.hip's kernel (lll_kernel<NQ, true>:
// lll.cpp:84-99, lll.h:125-140) in their own translation unit: the plain kernels keep their register budget, and
// the two files compile side by side.
#define FPHIP_LLL_KERNEL_EARLY 1
#define FPHIP_LLL_PROF 0
#include "lll_kernel.hip"
