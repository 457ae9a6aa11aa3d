// h264decode_amd/csrc/k_entropy_b.hip -- the B-slice build of the slice_data() kernel: k_entropy.hip compiled with two
// reference lists, Tables 7-14 / 7-18, direct prediction (8.4.1.2) and the B contexts of 9.3.  A separate kernel, so that
// the I/P kernel's registers, LDS and instruction cache footprint do not pay for it.
#define MI_ENT_B 1
#include "k_entropy.hip"
