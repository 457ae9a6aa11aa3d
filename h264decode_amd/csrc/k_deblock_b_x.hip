// h264decode_amd/csrc/k_deblock_b_x.hip -- the banded build of K5 for pictures with B slices (k_deblock_b_x).
#define MI_DB_B 1
#define MI_DB_BANDS 1
#include "k_deblock.hip"
