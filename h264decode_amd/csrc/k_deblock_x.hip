// h264decode_amd/csrc/k_deblock_x.hip -- K5 spread over several workgroups per picture (k_deblock_x): the banded build of k_deblock.hip,
// for launches with fewer pictures than the chip has CUs (see the header of k_deblock.hip).
#define MI_DB_BANDS 1
#include "k_deblock.hip"
