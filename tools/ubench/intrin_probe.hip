// What the byte / packed intrinsics K4 relies on really compute on this part (run on the GPU box: hipcc -O3 --offload-arch=gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s2 __attribute__((ext_vector_type(2)));
__global__ void probe(const uint32_t *in, uint32_t *out) {
    const uint32_t a = in[0], b = in[1], c = in[2];
    int k = 0;
    for (uint32_t sh = 0; sh < 4; sh++) out[k++] = __builtin_amdgcn_alignbyte(b, a, sh);
    out[k++] = __builtin_amdgcn_perm(b, a, 0x0c010c00u);
    out[k++] = __builtin_amdgcn_perm(b, a, 0x0c030c02u);
    out[k++] = __builtin_amdgcn_perm(b, a, 0x06040200u);
    out[k++] = __builtin_amdgcn_perm(b, a, 0x06050201u);
    out[k++] = __builtin_amdgcn_perm(0u, a, 0x0c030c02u);
    out[k++] = __builtin_amdgcn_lerp(a, b, 0x01010101u);
    out[k++] = static_cast<uint32_t>(__builtin_amdgcn_sdot4(static_cast<int>(a ^ 0x80808080u), 0x1414FB01, 4096, false));
    out[k++] = static_cast<uint32_t>(__builtin_amdgcn_sdot4(static_cast<int>(a ^ 0x80808080u), 0x000001FB, 0, false));
    out[k++] = __builtin_amdgcn_udot4(a, c, 32u, false);
    s2 x = __builtin_bit_cast(s2, a & 0x00FF00FFu), y = __builtin_bit_cast(s2, b & 0x00FF00FFu), z = __builtin_bit_cast(s2, c & 0x00FF00FFu);
    s2 v = (x + y) - (y + z) * static_cast<short>(5) + (x + z) * static_cast<short>(20);
    out[k++] = __builtin_bit_cast(uint32_t, v);
    const s2 zz = {0, 0}, mm = {255, 255};
    s2 cl = __builtin_elementwise_min(__builtin_elementwise_max((v + static_cast<short>(16)) >> 5, zz), mm);
    out[k++] = __builtin_bit_cast(uint32_t, cl);
    s2 neg = __builtin_bit_cast(s2, 0xFF00FFF0u); // (-16, -256)
    out[k++] = __builtin_bit_cast(uint32_t, (neg + static_cast<short>(16)) >> 5);
    // v_cvt_pk_i16_i32 narrows with saturation (K4's add_clip4 relies on it): (70000, -70000) -> 7fff 8000, (300, -300) -> 012c fed4
    out[k++] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(static_cast<int>(in[3]), -static_cast<int>(in[3])));
    out[k++] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(300, -300));
    out[k++] = __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(s2, 0x7FF00010u), __builtin_bit_cast(s2, static_cast<uint32_t>(in[3] & 0xFFu) * 0x00010001u))); // + (0x70, 0x70): 7fff 0080
}
int main() {
    uint32_t h[4] = {0x44332211u, 0x88776655u, 0x04030201u, 70000u}, *d_in, *d_out, o[32] = {0};
    hipMalloc(&d_in, 16), hipMalloc(&d_out, 128);
    hipMemcpy(d_in, h, 16, hipMemcpyHostToDevice);
    probe<<<1, 1>>>(d_in, d_out);
    hipMemcpy(o, d_out, 128, hipMemcpyDeviceToHost);
    const char *n[] = {"align0", "align1", "align2", "align3", "perm 0c010c00", "perm 0c030c02", "perm 06040200", "perm 06050201", "perm(0,a,0c030c02)", "lerp", "sdot4 T0", "sdot4 T1", "udot4", "pk tap", "pk clip", "pk ashr", "cvt_pk_i16 sat", "cvt_pk_i16", "pk add sat"};
    for (int i = 0; i < 19; i++) printf("%-20s %08x\n", n[i], o[i]);
    return 0;
}
