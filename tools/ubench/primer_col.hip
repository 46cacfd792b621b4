// Micro-benchmark of the primer-scan column (Myers HW step + bookkeeping + Eq fetch) in isolation, to compare
// instruction selections on gfx950 at the kernel's residency (4 workgroups of 4 waves per CU).
// Build: hipcc -O3 --offload-arch=gfx950 primer_col.hip -o primer_col ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int S = 80, CS = 84, NPs = 2, lNPs = 1;

template <int V>
__device__ __forceinline__ void step(unsigned Eq, unsigned &Pv, unsigned &Mv, int &score) {
    unsigned Xv = Eq | Mv;
    unsigned Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
    unsigned Ph = Mv | ~(Xh | Pv);
    unsigned Mh = Pv & Xh;
    score = score + (int)(Ph >> 31) + ((int)Mh >> 31);
    if (V == 1) {
        asm("v_add_u32 %0, %1, %1" : "=v"(Ph) : "v"(Ph));
        asm("v_add_u32 %0, %1, %1" : "=v"(Mh) : "v"(Mh));
    } else { Ph <<= 1; Mh <<= 1; }
    Pv = Mh | ~(Xv | Ph);
    Mv = Ph & Xv;
}

// V: 0 = as in the kernel, 1 = shifts as adds, 2 = no bookkeeping (core only), 3 = byte codes pre-scaled + ds_read_u8 per column,
//    4 = core + min only
template <int V>
__global__ __launch_bounds__(256, 4) void k(unsigned *out, int reps) {
    __shared__ unsigned char codes[128 * CS];
    __shared__ unsigned peq[16 * NPs];
    const int tid = threadIdx.x;
    for (int i = tid; i < 128 * CS; i += 256) codes[i] = (unsigned char)(((i * 2654435761u) >> 13) & 3) << (V == 3 ? lNPs + 2 : 0);
    for (int i = tid; i < 16 * NPs; i += 256) peq[i] = (i * 40503u) << 10;
    __syncthreads();
    const int p = tid & 1;
    const unsigned char *cw = codes + (tid >> 1) * CS;
    const unsigned *cw4 = (const unsigned *)cw;
    const unsigned *pq = peq + p;
    unsigned acc = 0;
    for (int rep = 0; rep < reps; rep++) {
        unsigned Pu = ~0u, Mu = acc;
        int sc = 20, bst = 21;
        unsigned gtw = 0, ltw = 0;
        if (V == 5 || V == 6) {   // plain byte codes, one ds_read_u8 per column (no packed-word extraction)
#pragma unroll 8
            for (int j = 0; j < S; j++) {
                unsigned e = pq[(unsigned)cw[j] << lNPs];
                step<(V == 6 ? 1 : 0)>(e, Pu, Mu, sc);
                ltw = __builtin_amdgcn_alignbit(ltw, (unsigned)(sc - bst), 31);
                gtw = __builtin_amdgcn_alignbit(gtw, (unsigned)(bst - sc), 31);
                bst = sc < bst ? sc : bst;
            }
        } else if (V == 3) {
            const char *pqb = (const char *)pq;
#pragma unroll 8
            for (int j = 0; j < S; j++) {
                unsigned e = *(const unsigned *)(pqb + cw[j]);
                step<0>(e, Pu, Mu, sc);
                ltw = __builtin_amdgcn_alignbit(ltw, (unsigned)(sc - bst), 31);
                gtw = __builtin_amdgcn_alignbit(gtw, (unsigned)(bst - sc), 31);
                bst = sc < bst ? sc : bst;
            }
        } else {
            unsigned cd = cw4[0];
            unsigned e0 = pq[(cd & 0xFF) << lNPs], e1 = pq[((cd >> 8) & 0xFF) << lNPs], e2 = pq[((cd >> 16) & 0xFF) << lNPs], e3 = pq[(cd >> 24) << lNPs];
            for (int g = 0; g < S / 4; g++) {
                int nx = g + 1 < S / 4 ? g + 1 : S / 4 - 1;
                cd = cw4[nx];
                unsigned n0 = pq[(cd & 0xFF) << lNPs], n1 = pq[((cd >> 8) & 0xFF) << lNPs], n2 = pq[((cd >> 16) & 0xFF) << lNPs], n3 = pq[(cd >> 24) << lNPs];
#define COL(E) do { step<V>(E, Pu, Mu, sc);                                                         \
                    if (V != 2 && V != 4) { ltw = __builtin_amdgcn_alignbit(ltw, (unsigned)(sc - bst), 31);        \
                                  gtw = __builtin_amdgcn_alignbit(gtw, (unsigned)(bst - sc), 31); }          \
                    if (V != 2) bst = sc < bst ? sc : bst; } while (0)
                COL(e0); COL(e1); COL(e2); COL(e3);
                e0 = n0; e1 = n1; e2 = n2; e3 = n3;
            }
        }
        acc ^= Pu ^ Mu ^ (unsigned)sc ^ (unsigned)bst ^ gtw ^ ltw;
    }
    out[blockIdx.x * 256 + tid] = acc;
}

template <int V>
void run(const char *name) {
    unsigned *out;
    const int blocks = 1024, reps = 2000;
    hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, reps);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves, each reps*S columns
    double ns_col_wave = ms * 1e6 / ((double)reps * S);
    printf("%-46s %.1f ns per column per wave = %.1f cycles of SIMD time per wave-column (4 waves/SIMD, 2.4 GHz)\n", name,
           ns_col_wave, ns_col_wave * 2.4 / 4.0);
    hipFree(out);
}

int main() {
    run<0>("V0 kernel's column");
    run<1>("V1 shifts as adds");
    run<2>("V2 core only (no min, no flags)");
    run<4>("V4 core + min");
    run<3>("V3 pre-scaled byte codes, ds_read_u8 per column");
    run<5>("V5 plain byte codes, ds_read_u8 per column");
    run<6>("V6 = V5 + shifts as adds");
    return 0;
}
