// Probe of ds_read_b64_tr_b16 semantics on gfx950 (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O2 tools/probes/tr16_probe.hip -o /tmp/tr16 && /tmp/tr16
// Lane t of each 16-lane group reads 4 bf16 at row (t>>2) + 4*group, cols 4*(t&3)..+3 of a [64][160] image
// holding S[r][c] = r*160 + c.  Expected (guide: out[l][j] = in[4j + ((l&15)>>2)][l&3]):
//   out[l][j] = S[4*(l>>4) + j][l & 15]
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ short S[64 * 160];
    for (int i = threadIdx.x; i < 64 * 160; i += 64) S[i] = (short)i;
    __syncthreads();
    int l = threadIdx.x;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(S + (l >> 2) * 160 + (l & 3) * 4));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    short* d; short h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) {
            int exp = (4 * (l >> 4) + j) * 160 + (l & 15);
            if (h[l * 4 + j] != exp) { if (bad < 8) printf("lane %d elem %d: got %d (row %d col %d) expected %d\n", l, j, h[l*4+j], h[l*4+j]/160, h[l*4+j]%160, exp); ++bad; }
        }
    printf("tr16 probe: %d mismatches\n", bad);
    return bad != 0;
}
