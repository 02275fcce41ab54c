// vmem_rate.hip - what does one vector-memory LOAD instruction cost on gfx950, by the shape of the addresses its 64 lanes form?
//
// The keypoint kernels of the front end (k_describe, k_select) read small 2-D windows (32 x 32 .. 37 x 40 bytes) at arbitrary positions
// of an image: few bytes per row, many rows.  This tool times the load shapes such a window can be read with - a byte per lane over 2
// rows, words over 7 rows x 36 B, 16-byte lanes over 16 rows, ... - at three residencies of the image (the window stays put: vector L1;
// random windows in 2 MiB: the XCD's L2; random windows in 1 GiB: HBM / Infinity Cache) with 16 waves per CU and 8 loads of a wave in
// flight, as the kernels run them.  Printed per shape and residency: ns per wave-instruction per CU (wall time x CUs / instructions) and
// the same in cycles at 2.4 GHz; rows = distinct image rows (>= cache lines) one instruction touches.
// Build + run on the GPU box:  hipcc -O2 --offload-arch=gfx950 tools/vmem_rate.hip -o /tmp/vmem_rate && /tmp/vmem_rate > gpurun_out/vmem_rate.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int PITCH = 704;           // level-0 pitch of the 640-px front end
constexpr int ITERS = 2048;          // windows per wave
constexpr int NB = 8;                // loads per window (in flight together)

enum { P_BYTE_2x32, P_BYTE_1x64, P_BYTE_GATHER, P_WORD_1x64, P_WORD_2x32, P_WORD_4x16, P_WORD_7x9, P_WORD_64x1, P_X2_8x8, P_X4_16x4, P_X4_4x16, P_COUNT };
static const char* NAMES[P_COUNT] = {"byte 2 rows x 32", "byte 1 row x 64", "byte gather in 37 x 37", "word 1 row x 64 (256 B)", "word 2 rows x 32 (128 B)", "word 4 rows x 16 (64 B)",
                                     "word 7 rows x 9 (36 B)", "word 64 rows x 1", "8-byte 8 rows x 8 (64 B)", "16-byte 16 rows x 4 (64 B)", "16-byte 4 rows x 16 (256 B)"};
static const int ROWS[P_COUNT] = {2, 1, 37, 1, 2, 4, 7, 64, 8, 16, 4};
static const int BYTES[P_COUNT] = {1, 1, 1, 4, 4, 4, 4, 4, 8, 16, 16};

template <int P> __host__ __device__ constexpr int bytes_of() { return P <= P_BYTE_GATHER ? 1 : P <= P_WORD_64x1 ? 4 : P == P_X2_8x8 ? 8 : 16; }

template <int P> __device__ __forceinline__ uint32_t lane_off(int lane, int k)
{
    switch (P) {
    case P_BYTE_2x32:   return ((lane >> 5) + 2 * k) * PITCH + (lane & 31);
    case P_BYTE_1x64:   return k * PITCH + lane;
    case P_BYTE_GATHER: return ((lane * 7 + k * 13) % 37) * PITCH + (lane * 11 + k * 5) % 37;
    case P_WORD_1x64:   return k * PITCH + 4 * lane;
    case P_WORD_2x32:   return ((lane >> 5) + 2 * k) * PITCH + 4 * (lane & 31);
    case P_WORD_4x16:   return ((lane >> 4) + 4 * k) * PITCH + 4 * (lane & 15);
    case P_WORD_7x9:    return (min(lane / 9, 6) + 7 * k) * PITCH + 4 * (lane % 9);
    case P_WORD_64x1:   return lane * PITCH + 4 * k;
    case P_X2_8x8:      return ((lane >> 3) + 8 * k) * PITCH + 8 * (lane & 7);
    case P_X4_16x4:     return ((lane >> 2) + 16 * k) * PITCH + 16 * (lane & 3);
    case P_X4_4x16:     return ((lane >> 4) + 4 * k) * PITCH + 16 * (lane & 15);
    }
    return 0;
}

// span = bytes of image the window origins are drawn from (0: the origin never moves).  The buffer is span + 256 rows long.
template <int P> __global__ __launch_bounds__(256) void k_load(const uint8_t* __restrict__ img, size_t span, uint32_t* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t off[NB];
#pragma unroll
    for (int k = 0; k < NB; k++) off[k] = lane_off<P>(lane, k);
    uint32_t rng = wave * 0x9e3779b9u + 12345u, acc = 0;
    constexpr int BY = bytes_of<P>();
    const size_t al = ~(size_t)(BY >= 4 ? 3 : 0);         // windows sit at any word (byte shapes: any byte) of a row; 8- and 16-byte lanes may be misaligned, as in an image
    for (int it = 0; it < ITERS; it++) {
        rng = rng * 1664525u + 1013904223u;
        const size_t o = span ? ((size_t)(((uint64_t)(rng >> 4) * (uint64_t)span) >> 28) & al) : 0;
        const uint8_t* w = img + o;                         // wave-uniform
        if (BY == 1) {
            uint32_t v[NB];
#pragma unroll
            for (int k = 0; k < NB; k++) v[k] = w[off[k]];
#pragma unroll
            for (int k = 0; k < NB; k++) acc += v[k];
        } else if (BY == 4) {
            uint32_t v[NB];
#pragma unroll
            for (int k = 0; k < NB; k++) v[k] = *reinterpret_cast<const uint32_t*>(w + off[k]);
#pragma unroll
            for (int k = 0; k < NB; k++) acc += v[k];
        } else if (BY == 8) {
            uint2 v[NB];
#pragma unroll
            for (int k = 0; k < NB; k++) v[k] = *reinterpret_cast<const uint2*>(w + off[k]);
#pragma unroll
            for (int k = 0; k < NB; k++) acc += v[k].x ^ v[k].y;
        } else {
            uint4 v[NB];
#pragma unroll
            for (int k = 0; k < NB; k++) v[k] = *reinterpret_cast<const uint4*>(w + off[k]);
#pragma unroll
            for (int k = 0; k < NB; k++) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
        }
    }
    if (acc == 0x12345u) out[wave] = acc;
}

template <int P> static int run(const uint8_t* img, uint32_t* out, int cus, bool last)
{
    const size_t spans[3] = {0, 2u << 20, 1u << 30};
    const char* res[3] = {"L1", "L2 (2 MiB)", "HBM (1 GiB)"};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("  {\"shape\": \"%s\", \"bytes_per_lane\": %d, \"rows_per_instr\": %d", NAMES[P], BYTES[P], ROWS[P]);
    for (int r = 0; r < 3; r++) {
        const int wgs = cus * 4;                            // 16 waves per CU
        k_load<P><<<wgs, 256>>>(img, spans[r], out);        // warm-up (and L2 fill)
        CK(hipEventRecord(e0));
        k_load<P><<<wgs, 256>>>(img, spans[r], out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double instr_per_cu = 16.0 * ITERS * NB, ns = ms * 1e6 / instr_per_cu;
        printf(", \"%s\": {\"ns_per_instr_per_cu\": %.2f, \"cycles\": %.1f, \"GB_per_s_chip\": %.0f}", res[r], ns, ns * 2.4, 64.0 * BYTES[P] * cus / ns);
    }
    printf("}%s\n", last ? "" : ",");
    return 0;
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint8_t* img; uint32_t* out;
    const size_t bytes = (1u << 30) + 1024 * PITCH;
    CK(hipMalloc(&img, bytes)); CK(hipMemset(img, 1, bytes)); CK(hipMalloc(&out, cus * 16 * 4));
    printf("{\"device\": \"%s\", \"cus\": %d, \"waves_per_cu\": 16, \"loads_in_flight_per_wave\": %d, \"pitch\": %d, \"shapes\": [\n", prop.gcnArchName, cus, NB, PITCH);
    if (run<P_BYTE_2x32>(img, out, cus, false) || run<P_BYTE_1x64>(img, out, cus, false) || run<P_BYTE_GATHER>(img, out, cus, false) ||
        run<P_WORD_1x64>(img, out, cus, false) || run<P_WORD_2x32>(img, out, cus, false) || run<P_WORD_4x16>(img, out, cus, false) ||
        run<P_WORD_7x9>(img, out, cus, false) || run<P_WORD_64x1>(img, out, cus, false) || run<P_X2_8x8>(img, out, cus, false) ||
        run<P_X4_16x4>(img, out, cus, false) || run<P_X4_4x16>(img, out, cus, true)) return 1;
    printf("]}\n");
    return 0;
}
