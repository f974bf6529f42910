// Host-side microbenchmark behind pk_field_upload_group_level's staging step: three separate float planes -> one array of
// {U,V,W} structs in a (pinned-like) destination, written with non-temporal stores.  Compares the scalar loop, the AVX2 shuffle
// form, thread counts, and a thread spawn per 256 MiB chunk against one persistent split.  Build + run:
//   /opt/rocm/lib/llvm/bin/clang++ -O3 -pthread tools/host_interleave_bench.cpp -o /tmp/hib && /tmp/hib
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static void scalar3(float* dst, const float* a, const float* b, const float* c, size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; i++) {
        __builtin_nontemporal_store(a[i], &dst[3 * i]);
        __builtin_nontemporal_store(b[i], &dst[3 * i + 1]);
        __builtin_nontemporal_store(c[i], &dst[3 * i + 2]);
    }
}
typedef float v8f __attribute__((vector_size(32)));
typedef float v8fu __attribute__((vector_size(32), aligned(4)));
typedef float v16f __attribute__((vector_size(64)));
__attribute__((target("avx2"))) static void vec3f(float* dst, const float* a, const float* b, const float* c, size_t lo, size_t hi) {
    size_t i = lo;
    for (; i + 8 <= hi; i += 8) {
        v8f va = *(const v8fu*)(a + i), vb = *(const v8fu*)(b + i), vc = *(const v8fu*)(c + i);
        v16f ab = __builtin_shufflevector(va, vb, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
        v16f cc = __builtin_shufflevector(vc, vc, 0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3, 4, 5, 6, 7);
        v8f o0 = __builtin_shufflevector(ab, cc, 0, 8, 16, 1, 9, 17, 2, 10);
        v8f o1 = __builtin_shufflevector(ab, cc, 18, 3, 11, 19, 4, 12, 20, 5);
        v8f o2 = __builtin_shufflevector(ab, cc, 13, 21, 6, 14, 22, 7, 15, 23);
        __builtin_nontemporal_store(o0, (v8f*)(dst + 3 * i));
        __builtin_nontemporal_store(o1, (v8f*)(dst + 3 * i + 8));
        __builtin_nontemporal_store(o2, (v8f*)(dst + 3 * i + 16));
    }
    scalar3(dst, a, b, c, i, hi);
    __builtin_ia32_sfence();
}

int main(int argc, char** argv) {
    const size_t n = (size_t)256 << 20;  // elements per plane: 1 GiB each, 3 GiB out
    float *a, *b, *c, *d;
    posix_memalign((void**)&a, 4096, n * 4);
    posix_memalign((void**)&b, 4096, n * 4);
    posix_memalign((void**)&c, 4096, n * 4);
    posix_memalign((void**)&d, 4096, n * 12);
    {  // first touch in parallel, like a multi-threaded producer would
        std::vector<std::thread> pool;
        for (int k = 0; k < 16; k++)
            pool.emplace_back([=]() {
                for (size_t i = k * (n / 16); i < (k + 1) * (n / 16); i++) { a[i] = (float)i; b[i] = 2.f * i; c[i] = 3.f * i; }
                memset(d + 3 * k * (n / 16), 0, 12 * (n / 16));
            });
        for (auto& t : pool) t.join();
    }
    const size_t chunk = ((size_t)256 << 20) / 12 / 8 * 8;
    for (int nthr : {8, 16, 32, 64, 128})
        for (int mode = 0; mode < 2; mode++)
            for (int chunked = 0; chunked < 2; chunked++) {
                double best = 0;
                for (int rep = 0; rep < 2; rep++) {
                    auto t0 = std::chrono::steady_clock::now();
                    const size_t step = chunked ? chunk : n;
                    for (size_t base = 0; base < n; base += step) {
                        const size_t end = std::min(n, base + step);
                        std::vector<std::thread> pool;
                        const size_t per = (((end - base) + nthr - 1) / nthr + 7) & ~(size_t)7;
                        for (int k = 0; k < nthr; k++) {
                            const size_t lo = base + k * per, hi = std::min(end, lo + per);
                            if (lo >= hi) break;
                            pool.emplace_back([=]() { mode ? vec3f(d, a, b, c, lo, hi) : scalar3(d, a, b, c, lo, hi); });
                        }
                        for (auto& t : pool) t.join();
                    }
                    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    best = std::max(best, n * 12 / s / 1e9);
                }
                printf("threads %3d  %s  %s : %7.1f GB/s out\n", nthr, mode ? "avx2  " : "scalar", chunked ? "spawn per 256MiB" : "one split       ", best);
                fflush(stdout);
            }
    for (size_t i = 0; i < n; i += 9973)
        if (d[3 * i] != a[i] || d[3 * i + 1] != b[i] || d[3 * i + 2] != c[i]) { printf("MISMATCH %zu\n", i); return 1; }
    return 0;
}
