// Host side of the field-level stream: pageable NumPy memory -> pinned staging chunk, split over a persistent pool of host threads.
// Plain C++ (no HIP): compiled for the host only, so the x86 target attributes below never meet the gfx950 pass.
//
// What it replaces in the reference: WindowedArray reads a level and hands NumPy the array (src/parcels/_core/_windowed_array.py:
// 56-97); here the level has to cross PCIe, and the staging fill must outrun the link (57 GB/s measured, profiles/r02_c_pcie_
// ceiling.json) with room to spare because the DMA engine reads the same DRAM.  Measured on the bench box (2 x EPYC 9575F), three
// float planes -> {U,V,W} structs, tools/host_interleave_bench.cpp: scalar loop + a thread spawn per chunk 86 GB/s; AVX2 shuffles +
// one persistent split 269 GB/s.
#include "pk_host_stage.h"

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <unistd.h>

namespace pkhost {

unsigned copy_threads() {
    static const unsigned n = [] {
        if (const char* e = getenv("PK_COPY_THREADS")) return (unsigned)std::max(1, atoi(e));
        const unsigned hw = std::thread::hardware_concurrency();
        return std::max(8u, std::min(32u, hw / 4));
    }();
    return n;
}

// ---- persistent pool: run(n, f) executes f(0..n-1), the caller takes index 0 ---------------------------------------------------
class Pool {
    std::vector<std::thread> workers;
    std::mutex m, run_m;
    std::condition_variable go, done;
    const std::function<void(unsigned)>* job = nullptr;
    unsigned njob = 0, pending = 0;
    unsigned long generation = 0;
    bool stop = false;
    pid_t owner = getpid();

    void worker(unsigned index) {
        unsigned long seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            go.wait(lk, [&] { return stop || generation != seen; });
            if (stop) return;
            seen = generation;
            if (index < njob) {
                const auto* f = job;
                lk.unlock();
                (*f)(index);
                lk.lock();
                if (--pending == 0) done.notify_one();
            }
        }
    }

  public:
    void run(unsigned n, const std::function<void(unsigned)>& f) {
        if (n <= 1) {
            f(0);
            return;
        }
        std::lock_guard<std::mutex> serial(run_m);  // one job at a time (two contexts may stream from two Python threads)
        {
            std::lock_guard<std::mutex> lk(m);
            if (owner != getpid()) {
                // a fork()ed child (multiprocessing) inherits the bookkeeping but none of the threads: start over.  The thread
                // objects of the parent's workers cannot be destroyed here (joinable), so they are parked for the process lifetime.
                if (!workers.empty()) new std::vector<std::thread>(std::move(workers));
                workers.clear();
                owner = getpid();
            }
            while (workers.size() + 1 < n) {
                const unsigned index = (unsigned)workers.size() + 1;
                workers.emplace_back([this, index] { worker(index); });
            }
            job = &f;
            njob = n;
            pending = n - 1;
            generation++;
        }
        go.notify_all();
        f(0);
        std::unique_lock<std::mutex> lk(m);
        done.wait(lk, [&] { return pending == 0; });
        job = nullptr;
    }
    ~Pool() {
        if (owner != getpid()) {  // forked child that never streamed: nothing of ours to join
            if (!workers.empty()) new std::vector<std::thread>(std::move(workers));
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        go.notify_all();
        for (auto& t : workers) t.join();
    }
};
static Pool& pool() {
    static Pool p;
    return p;
}

void parallel_memcpy(void* dst, const void* src, size_t bytes) {
    const size_t min_chunk = 4u << 20;
    const unsigned nthr = (unsigned)std::min<size_t>(copy_threads(), std::max<size_t>(1, bytes / min_chunk));
    const size_t chunk = ((bytes / nthr) + 4095) & ~(size_t)4095;
    pool().run(nthr, [=](unsigned k) {
        const size_t off = (size_t)k * chunk;
        if (off < bytes) memcpy((char*)dst + off, (const char*)src + off, std::min(chunk, bytes - off));
    });
}

// ---- interleave: elements [lo, hi) of `ncomp` planes -> array of structs, non-temporal stores ---------------------------------
// The pinned chunk is written once and read by the DMA engine, never by this core: no read-for-ownership of the destination lines.
template <class T>
static void interleave_scalar(T* dst, const T* const* src, int ncomp, size_t lo, size_t hi) {
    if (ncomp == 3) {
        const T *a = src[0], *b = src[1], *c = src[2];
        for (size_t i = lo; i < hi; i++) {
            __builtin_nontemporal_store(a[i], &dst[3 * i]);
            __builtin_nontemporal_store(b[i], &dst[3 * i + 1]);
            __builtin_nontemporal_store(c[i], &dst[3 * i + 2]);
        }
    } else if (ncomp == 2) {
        const T *a = src[0], *b = src[1];
        for (size_t i = lo; i < hi; i++) {
            __builtin_nontemporal_store(a[i], &dst[2 * i]);
            __builtin_nontemporal_store(b[i], &dst[2 * i + 1]);
        }
    } else {
        for (size_t i = lo; i < hi; i++)
            for (int k = 0; k < ncomp; k++) dst[(size_t)ncomp * i + k] = src[k][i];
    }
}

typedef float v8f __attribute__((vector_size(32)));
typedef float v8fu __attribute__((vector_size(32), aligned(4)));
typedef float v16f __attribute__((vector_size(64)));
typedef double v4d __attribute__((vector_size(32)));
typedef double v4du __attribute__((vector_size(32), aligned(8)));
typedef double v8d __attribute__((vector_size(64)));

// 8 (float) / 4 (double) elements per plane per iteration -> three (two) full 32-byte non-temporal stores; `lo` is a multiple of
// the vector width and dst is page-aligned, so every store is 32-byte aligned
__attribute__((target("avx2"))) static void interleave3_avx2(float* dst, const float* a, const float* b, const float* c, size_t lo, size_t hi) {
    size_t i = lo;
    for (; i + 8 <= hi; i += 8) {
        const v8f va = *(const v8fu*)(a + i), vb = *(const v8fu*)(b + i), vc = *(const v8fu*)(c + i);
        const v16f ab = __builtin_shufflevector(va, vb, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
        const v16f cc = __builtin_shufflevector(vc, vc, 0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3, 4, 5, 6, 7);
        __builtin_nontemporal_store((v8f)__builtin_shufflevector(ab, cc, 0, 8, 16, 1, 9, 17, 2, 10), (v8f*)(dst + 3 * i));
        __builtin_nontemporal_store((v8f)__builtin_shufflevector(ab, cc, 18, 3, 11, 19, 4, 12, 20, 5), (v8f*)(dst + 3 * i + 8));
        __builtin_nontemporal_store((v8f)__builtin_shufflevector(ab, cc, 13, 21, 6, 14, 22, 7, 15, 23), (v8f*)(dst + 3 * i + 16));
    }
    const float* src[3] = {a, b, c};
    interleave_scalar(dst, src, 3, i, hi);
}
__attribute__((target("avx2"))) static void interleave3_avx2(double* dst, const double* a, const double* b, const double* c, size_t lo, size_t hi) {
    size_t i = lo;
    for (; i + 4 <= hi; i += 4) {
        const v4d va = *(const v4du*)(a + i), vb = *(const v4du*)(b + i), vc = *(const v4du*)(c + i);
        const v8d ab = __builtin_shufflevector(va, vb, 0, 1, 2, 3, 4, 5, 6, 7);
        const v8d cc = __builtin_shufflevector(vc, vc, 0, 1, 2, 3, 0, 1, 2, 3);
        __builtin_nontemporal_store((v4d)__builtin_shufflevector(ab, cc, 0, 4, 8, 1), (v4d*)(dst + 3 * i));
        __builtin_nontemporal_store((v4d)__builtin_shufflevector(ab, cc, 5, 9, 2, 6), (v4d*)(dst + 3 * i + 4));
        __builtin_nontemporal_store((v4d)__builtin_shufflevector(ab, cc, 10, 3, 7, 11), (v4d*)(dst + 3 * i + 8));
    }
    const double* src[3] = {a, b, c};
    interleave_scalar(dst, src, 3, i, hi);
}
__attribute__((target("avx2"))) static void interleave2_avx2(float* dst, const float* a, const float* b, size_t lo, size_t hi) {
    size_t i = lo;
    for (; i + 8 <= hi; i += 8) {
        const v8f va = *(const v8fu*)(a + i), vb = *(const v8fu*)(b + i);
        __builtin_nontemporal_store((v8f)__builtin_shufflevector(va, vb, 0, 8, 1, 9, 2, 10, 3, 11), (v8f*)(dst + 2 * i));
        __builtin_nontemporal_store((v8f)__builtin_shufflevector(va, vb, 4, 12, 5, 13, 6, 14, 7, 15), (v8f*)(dst + 2 * i + 8));
    }
    const float* src[2] = {a, b};
    interleave_scalar(dst, src, 2, i, hi);
}
__attribute__((target("avx2"))) static void interleave2_avx2(double* dst, const double* a, const double* b, size_t lo, size_t hi) {
    size_t i = lo;
    for (; i + 4 <= hi; i += 4) {
        const v4d va = *(const v4du*)(a + i), vb = *(const v4du*)(b + i);
        __builtin_nontemporal_store((v4d)__builtin_shufflevector(va, vb, 0, 4, 1, 5), (v4d*)(dst + 2 * i));
        __builtin_nontemporal_store((v4d)__builtin_shufflevector(va, vb, 2, 6, 3, 7), (v4d*)(dst + 2 * i + 4));
    }
    const double* src[2] = {a, b};
    interleave_scalar(dst, src, 2, i, hi);
}

static bool have_avx2() {
    static const bool v = [] {
        if (const char* e = getenv("PK_STAGE_SCALAR")) return atoi(e) == 0;
        return (bool)__builtin_cpu_supports("avx2");
    }();
    return v;
}

template <class T>
static void interleave_range(T* dst, const T* const* src, int ncomp, size_t lo, size_t hi) {
    const bool aligned = ((uintptr_t)dst & 31) == 0;  // + lo a multiple of 8: see parallel_interleave
    if (aligned && have_avx2() && ncomp == 3) interleave3_avx2(dst, src[0], src[1], src[2], lo, hi);
    else if (aligned && have_avx2() && ncomp == 2) interleave2_avx2(dst, src[0], src[1], lo, hi);
    else interleave_scalar(dst, src, ncomp, lo, hi);
    __builtin_ia32_sfence();  // the weakly-ordered stores are globally visible before the DMA is queued
}

template <class T>
void parallel_interleave(T* dst, const T* const* src, int ncomp, size_t n) {
    const size_t min_chunk = (size_t)1 << 18;
    const unsigned nthr = (unsigned)std::min<size_t>(copy_threads(), std::max<size_t>(1, n / min_chunk));
    const size_t per = (((n + nthr - 1) / nthr) + 7) & ~(size_t)7;  // every thread starts on a 32-byte boundary of dst
    pool().run(nthr, [=](unsigned k) {
        const size_t lo = (size_t)k * per, hi = std::min(n, lo + per);
        if (lo < hi) interleave_range(dst, src, ncomp, lo, hi);
    });
}
template void parallel_interleave<float>(float*, const float* const*, int, size_t);
template void parallel_interleave<double>(double*, const double* const*, int, size_t);

// every interleave variant (float / double, 2 / 3 planes, vector body + scalar tail, one thread / the pool) against the plain loop
template <class T>
static int selftest_type() {
    int bad = 0;
    for (size_t n : {(size_t)1, (size_t)7, (size_t)8, (size_t)9, (size_t)1000, (size_t)300007, ((size_t)1 << 20) + 13}) {
        for (int ncomp = 2; ncomp <= 3; ncomp++) {
            std::vector<T> planes[3];
            const T* src[3];
            for (int k = 0; k < 3; k++) {
                planes[k].resize(n + 1);
                for (size_t i = 0; i <= n; i++) planes[k][i] = (T)((double)(i * 3 + (size_t)k) * 0.37 - 11.0);
                src[k] = planes[k].data() + 1;  // deliberately not 32-byte aligned
            }
            void* p = nullptr;
            if (posix_memalign(&p, 4096, n * (size_t)ncomp * sizeof(T) + 64)) return 1000;
            T* dst = (T*)p;
            parallel_interleave(dst, src, ncomp, n);
            for (size_t i = 0; i < n && !bad; i++)
                for (int k = 0; k < ncomp; k++)
                    if (dst[(size_t)ncomp * i + k] != src[k][i]) { bad++; break; }
            free(p);
        }
    }
    return bad;
}
}  // namespace pkhost

extern "C" int32_t pk_host_stage_selftest(void) { return pkhost::selftest_type<float>() + pkhost::selftest_type<double>(); }

namespace pkhost {

}  // namespace pkhost
