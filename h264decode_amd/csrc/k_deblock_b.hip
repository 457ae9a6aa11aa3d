// h264decode_amd/csrc/k_deblock_b.hip -- K5 for pictures with B slices: k_deblock.hip compiled with the two-list boundary
// strength rule of 8.7.2.1 (a separate kernel: the I/P pictures' kernel keeps its registers and LDS).
#define MI_DB_B 1
#include "k_deblock.hip"
