// Probe: what does a CU-masked HIP stream (hipExtStreamCreateWithCUMask) do on MI355X?
//   * which bit of the mask enables which (XCC, SE, CU): every workgroup records HW_REG_XCC_ID and HW_REG_HW_ID;
//   * does a mask that leaves whole XCDs without CUs still run every workgroup (and where);
//   * does the observed "block b runs on XCD b % 8" rule survive a mask;
//   * streaming-copy and MFMA rates of a masked stream alone and of four masked streams side by side.
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_cumask tools/probes/probe_cumask.hip && timeout 120 /tmp/probe_cumask
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

// s_getreg_b32 simm16 = (size-1) << 11 | offset << 6 | id;  HW_REG_HW_ID = 4, HW_REG_XCC_ID = 20 (gfx940+)
__global__ void where_kernel(uint32_t* out, int spin) {
    uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
    uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    // keep the workgroup resident for a while so that a grid spreads over every enabled CU
    uint64_t t0 = __builtin_readcyclecounter();
    while ((int64_t)(__builtin_readcyclecounter() - t0) < spin) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}

__global__ void copy_kernel(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) b[i] = a[i];
}

typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) float f4;
__global__ void __launch_bounds__(256) mfma_kernel(float* out, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.f) out[0] = 1.f;
}

static hipStream_t masked_stream(const std::vector<uint32_t>& mask) {
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    return s;
}

static void census(const char* name, hipStream_t s, int blocks, uint32_t* d_out, std::vector<uint32_t>& h) {
    CK(hipMemsetAsync(d_out, 0xff, sizeof(uint32_t) * 2 * blocks, s));
    where_kernel<<<blocks, 64, 0, s>>>(d_out, 200000);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d_out, sizeof(uint32_t) * 2 * blocks, hipMemcpyDeviceToHost));
    int per_xcc[16] = {0};
    bool seen[16][8][2][16];
    memset(seen, 0, sizeof(seen));
    int mod_ok = 0;
    for (int b = 0; b < blocks; b++) {
        uint32_t xcc = h[2 * b] & 15, hw = h[2 * b + 1];
        uint32_t cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_xcc[xcc]++;
        seen[xcc][se][sh][cu] = true;
        mod_ok += ((uint32_t)(b % 8) == xcc);
    }
    printf("%-28s blocks %5d  per-XCC:", name, blocks);
    int total_cus = 0;
    for (int x = 0; x < 8; x++) {
        int cus = 0;
        for (int se = 0; se < 8; se++) for (int sh = 0; sh < 2; sh++) for (int cu = 0; cu < 16; cu++) cus += seen[x][se][sh][cu];
        total_cus += cus;
        printf(" %d:%d wg/%d cu", x, per_xcc[x], cus);
    }
    printf("  | distinct CUs %d, (block %% 8 == xcc) for %d of %d\n", total_cus, mod_ok, blocks);
}

static double time_copy(hipStream_t s, const float4* a, float4* b, size_t n, int grid) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    copy_kernel<<<grid, 256, 0, s>>>(a, b, n);
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 5; i++) copy_kernel<<<grid, 256, 0, s>>>(a, b, n);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return 5.0 * 2.0 * n * 16 / (ms * 1e-3) / 1e12;  // TB/s read+write
}

static double time_mfma(hipStream_t s, float* out, int grid, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    mfma_kernel<<<grid, 256, 0, s>>>(out, iters);
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 3; i++) mfma_kernel<<<grid, 256, 0, s>>>(out, iters);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return 3.0 * grid * 4.0 * iters * 4.0 * 2.0 * 16 * 16 * 32 / (ms * 1e-3) / 1e12;  // TFLOP/s
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s, %d CUs\n", p.name, p.multiProcessorCount);
    const int NB = 2048;
    uint32_t* d_out; CK(hipMalloc(&d_out, sizeof(uint32_t) * 2 * NB));
    std::vector<uint32_t> h(2 * NB);
    hipStream_t plain; CK(hipStreamCreate(&plain));
    census("plain stream", plain, NB, d_out, h);

    // masks: 256 bits = 8 words
    auto mk = [](auto pred) { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 256; i++) if (pred(i)) m[i / 32] |= 1u << (i % 32); return m; };
    struct { const char* name; std::vector<uint32_t> m; } masks[] = {
        {"bits i%8 in {0,1}", mk([](int i) { return i % 8 < 2; })},
        {"bits i%8 in {2,3}", mk([](int i) { return i % 8 == 2 || i % 8 == 3; })},
        {"bits i%8 == 5", mk([](int i) { return i % 8 == 5; })},
        {"bits 0..63", mk([](int i) { return i < 64; })},
        {"bits 64..127", mk([](int i) { return i >= 64 && i < 128; })},
        {"bits (i/8)%4 == 0", mk([](int i) { return (i / 8) % 4 == 0; })},
        {"bits 0..127", mk([](int i) { return i < 128; })},
        {"bits i%8 < 4", mk([](int i) { return i % 8 < 4; })},
        {"bit 0 only", mk([](int i) { return i == 0; })},
        {"bit 9 only", mk([](int i) { return i == 9; })},
    };
    std::vector<hipStream_t> st;
    for (auto& m : masks) {
        hipStream_t s = masked_stream(m.m);
        st.push_back(s);
        census(m.name, s, NB, d_out, h);
        census(m.name, s, 64, d_out, h);
    }

    // rates: copy 256 MB and MFMA on the plain stream and on masked streams
    size_t n = (size_t)16 << 20;  // float4 elements = 256 MiB
    float4 *a, *b; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16));
    CK(hipMemset(a, 1, n * 16));
    float* mo; CK(hipMalloc(&mo, 64));
    printf("copy  TB/s (r+w): plain %.2f | xcd{0,1} %.2f | bits0..63 %.2f | (i/8)%%4==0 %.2f | half(i%%8<4) %.2f\n",
           time_copy(plain, a, b, n, 2048), time_copy(st[0], a, b, n, 2048), time_copy(st[3], a, b, n, 2048), time_copy(st[5], a, b, n, 2048),
           time_copy(st[7], a, b, n, 2048));
    printf("mfma TFLOP/s   : plain %.0f | xcd{0,1} %.0f | bits0..63 %.0f | (i/8)%%4==0 %.0f | half(i%%8<4) %.0f\n",
           time_mfma(plain, mo, 2048, 4000), time_mfma(st[0], mo, 2048, 4000), time_mfma(st[3], mo, 2048, 4000), time_mfma(st[5], mo, 2048, 4000),
           time_mfma(st[7], mo, 2048, 4000));

    // four disjoint 2-XCD streams side by side vs four plain streams side by side (MFMA kernels)
    std::vector<hipStream_t> quad, quadp;
    for (int q = 0; q < 4; q++) {
        quad.push_back(masked_stream(mk([q](int i) { return (i % 8) / 2 == q; })));
        hipStream_t s; CK(hipStreamCreate(&s)); quadp.push_back(s);
    }
    for (int which = 0; which < 2; which++) {
        auto& ss = which ? quadp : quad;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 4; r++) for (auto s : ss) mfma_kernel<<<512, 256, 0, s>>>(mo, 4000);
        for (auto s : ss) CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("4 streams x 4 MFMA launches (%s): %.3f ms -> %.0f TFLOP/s aggregate\n", which ? "plain" : "2 XCDs each", ms,
               16.0 * 512 * 4.0 * 4000 * 4.0 * 2.0 * 16 * 16 * 32 / (ms * 1e-3) / 1e12);
    }
    for (int q = 0; q < 4; q++) { char nm[64]; snprintf(nm, sizeof nm, "quad stream %d", q); census(nm, quad[q], 512, d_out, h); }
    printf("done\n");
    return 0;
}
