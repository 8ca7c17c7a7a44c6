// Dev tool: calibrates rocprofv3 FETCH_SIZE on gfx950 for the two access shapes of k_spatial:
//   stream64 : every lane reads one 64-B record with 4 x dwordx4 (record stride 64 B, in order)
//   gather64 : every lane reads one 64-B record at a RANDOM index of a 1 GiB table (4 x dwordx4)
// Known bytes: n_records * 64. Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void stream64(const float4* __restrict__ t, size_t n, float* out)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 a = t[4 * i], b = t[4 * i + 1], c = t[4 * i + 2], d = t[4 * i + 3];
    float s = a.x + b.y + c.z + d.w;
    if (s == 12345.678f) out[0] = s;
}
__global__ void gather64(const float4* __restrict__ t, size_t n_table, size_t n, float* out)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t h = i * 6364136223846793005ULL + 1442695040888963407ULL;
    h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ULL; h ^= h >> 32;
    size_t j = h % n_table;
    float4 a = t[4 * j], b = t[4 * j + 1], c = t[4 * j + 2], d = t[4 * j + 3];
    float s = a.x + b.y + c.z + d.w;
    if (s == 12345.678f) out[0] = s;
}
int main()
{
    const size_t n_table = (size_t)1 << 24; // 16M records x 64 B = 1 GiB
    float4* t; float* out;
    hipMalloc(&t, n_table * 64); hipMalloc(&out, 4);
    hipMemset(t, 0, n_table * 64);
    const size_t n_stream = n_table, n_gather = (size_t)1 << 22; // 4M gathers = 256 MiB of records
    for (int rep = 0; rep < 3; ++rep)
    {
        stream64<<<(unsigned)((n_stream + 255) / 256), 256>>>(t, n_stream, out);
        gather64<<<(unsigned)((n_gather + 255) / 256), 256>>>(t, n_table, n_gather, out);
    }
    hipDeviceSynchronize();
    printf("stream64 known bytes %zu, gather64 known bytes %zu\n", n_stream * 64, n_gather * 64);
    return 0;
}
