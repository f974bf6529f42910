// Host-only helpers of the field-level stream (pk_host_stage.cpp): pageable -> pinned staging fills on a persistent thread pool.
#pragma once
#include <cstddef>
#include <cstdint>

namespace pkhost {
unsigned copy_threads();  // PK_COPY_THREADS, default min(32, hardware threads / 4), at least 8
void parallel_memcpy(void* dst, const void* src, size_t bytes);
// elements [0, n) of `ncomp` separate host planes -> one array of structs {c0, c1, ...} at dst (32-byte aligned), non-temporal stores
template <class T>
void parallel_interleave(T* dst, const T* const* src, int ncomp, size_t n);
}  // namespace pkhost
