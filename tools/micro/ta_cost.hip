// Microbenchmark: what a divergent global_load_dwordx4 costs on the vector-memory path of gfx950, by lane pattern.
// Records of 64 B in an L2-resident table (the size of cornellbox's pair array); every CU runs 32 waves.
// build: hipcc --offload-arch=gfx950 -O3 -o build/ta_cost tools/micro/ta_cost.hip ; run: build/ta_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// mode 0: every active lane its own record, ONE 16-byte load (piece 0)
// mode 1: every active lane its own record, FOUR 16-byte loads (the whole record: what the walks do)
// mode 2: every QUAD one record, lane i of the quad loads piece i: one load instruction fetches the record for one lane of the quad
// mode 3: every active lane its own record, one 4-byte load
// active: lane mask pattern 0 = all 64, 1 = lanes 0..7, 2 = one lane per 8 (0, 8, 16, ..), 3 = lanes 0..15, 4 = one lane per quad (16 lanes)
template <int MODE>
__global__ __launch_bounds__(1024) void k(const f4* __restrict__ tab, uint32_t n_rec, int iters, int active, float* out) {
    const uint32_t lane = threadIdx.x & 63;
    bool on = true;
    if (active == 1) on = lane < 8;
    if (active == 2) on = (lane & 7) == 0;
    if (active == 3) on = lane < 16;
    if (active == 4) on = (lane & 3) == 0;
    uint32_t st = mix(blockIdx.x * 1024 + threadIdx.x + 1);
    f4 acc = {0, 0, 0, 0};
    if (on) {
        for (int i = 0; i < iters; i++) {
            st = mix(st + 0x9e3779b9u);
            uint32_t rec = st % n_rec;
            if (MODE == 2) rec = __shfl(rec, lane & ~3u); // the quad's record (the shuffle is a DPP move)
            const f4* p = tab + (size_t)rec * 4;
            if (MODE == 0) acc += p[0];
            if (MODE == 1) { acc += p[0]; acc += p[1]; acc += p[2]; acc += p[3]; }
            if (MODE == 2) acc += p[lane & 3];
            if (MODE == 3) acc.x += ((const float*)p)[0];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;
}

int main(int argc, char** argv) {
    const uint32_t n_rec = argc > 1 ? (uint32_t)atoi(argv[1]) : 13169; // 13169 = cornellbox's pair array (843 KB, L2-resident); 256 = 16 KB, L1-resident
    std::vector<float> h((size_t)n_rec * 16, 1.f);
    f4* d; float* out;
    hipMalloc(&d, h.size() * 4); hipMalloc(&out, 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("table: %u records of 64 B = %.0f KB\n", n_rec, n_rec * 64 / 1024.0);
    const int grid = cus * 2, iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const char* mname[4] = {"1 x 16 B per lane", "4 x 16 B per lane (whole record)", "quad-cooperative 16 B (1 record per quad)", "1 x 4 B per lane"};
    const char* aname[5] = {"64 lanes", "lanes 0..7", "8 lanes, one per 8", "lanes 0..15", "16 lanes, one per quad"};
    const int alanes[5] = {64, 8, 8, 16, 16};
    for (int mode = 0; mode < 4; mode++)
        for (int act = 0; act < 5; act++) {
            if (mode == 2 && act != 0) continue;
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(1024), 0, 0, d, n_rec, iters, act, out);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(1024), 0, 0, d, n_rec, iters, act, out);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(1024), 0, 0, d, n_rec, iters, act, out);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(1024), 0, 0, d, n_rec, iters, act, out);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            const double loads_per_iter = mode == 1 ? 4 : 1;
            const double wave_instr_per_cu = 32.0 * iters * loads_per_iter; // 32 waves per CU
            const double ns_per_instr = best * 1e6 / wave_instr_per_cu;
            printf("%-44s %-24s %8.3f ms  %7.2f ns per wave-instruction per CU (%6.1f cyc at 2.4 GHz)  %6.2f cyc per active lane-load\n", mname[mode], aname[act], best,
                   ns_per_instr, ns_per_instr * 2.4, ns_per_instr * 2.4 / alanes[act]);
        }
    return 0;
}
