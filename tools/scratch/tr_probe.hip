// what ds_read_b64_tr_b16 returns: LDS holds lds[i] = i (u16); lane l supplies element address addr[l]; prints the 4 values per lane
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const int* __restrict__ addr_elems, unsigned short* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int a = addr_elems[threadIdx.x];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(&lds[a]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
    int h_addr[64]; unsigned short h_out[256];
    int *d_addr; unsigned short* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    for (int pat = 0; pat < 2; ++pat) {
        const int VLD = 160;
        for (int l = 0; l < 64; ++l) {
            if (pat == 0) h_addr[l] = 4 * l;     // contiguous 4 x 16 block per 16-lane group
            else { int hi = l >> 5; h_addr[l] = (4 * hi + ((l & 15) >> 2)) * VLD + ((l >> 4) & 1) * 16 + (l & 3) * 4; }   // attention_wide's tr_off
        }
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("pattern %d (element = row * %d + col for pattern 1)\n", pat, VLD);
        int bad = 0;
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d addr %5d ->", l, h_addr[l]);
            for (int j = 0; j < 4; ++j) {
                int e = h_out[l * 4 + j];
                if (pat == 0) { printf(" %5d", e); if (e != (l & 15) + j * 16 + (l >> 4) * 64) ++bad; }
                else { printf(" (r%2d,c%2d)", e / VLD, e % VLD); int hi = l >> 5; if (e / VLD != 4 * hi + j || e % VLD != (l & 31)) ++bad; }
            }
            printf("\n");
        }
        printf("pattern %d mismatches vs expectation: %d\n", pat, bad);
    }
    return 0;
}
