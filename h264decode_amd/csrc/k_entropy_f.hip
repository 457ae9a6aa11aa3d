// h264decode_amd/csrc/k_entropy_f.hip -- the slice-group build of the I/P slice_data() kernel: k_entropy.hip compiled with the
// macroblock walk of 8.2.2 (nextMbAddress over the picture's mbToSliceGroupMap; h264/slice.go:134-158, :530-552) and neighbour
// entries validated by row.  A separate kernel: only launches that hold a picture with more than one slice group use it.
#define MI_ENT_FMO 1
#include "k_entropy.hip"
