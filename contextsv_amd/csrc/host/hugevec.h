// hugevec.h — growable array on anonymous mmap with transparent huge pages requested (MADV_HUGEPAGE) and growth by mremap.
// The BAM decode path fills hundreds of megabytes of fresh memory per contig; with 4 KiB pages the page faults cost more
// than the inflate (measured: 1.5-4 s of a chr22 30x decode), and std::vector growth re-faults everything on every doubling.
#pragma once
#include <sys/mman.h>

#include <cstddef>
#include <cstring>
#include <new>
#include <type_traits>
#include <utility>

template <class T>
class HugeVec {
    static_assert(std::is_trivially_copyable<T>::value, "HugeVec holds plain data");
public:
    HugeVec() = default;
    ~HugeVec() { release(); }
    HugeVec(const HugeVec &) = delete;
    HugeVec &operator=(const HugeVec &) = delete;
    HugeVec(HugeVec &&o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
    HugeVec &operator=(HugeVec &&o) noexcept
    {
        if (this != &o) { release(); p = o.p; n = o.n; cap = o.cap; o.p = nullptr; o.n = o.cap = 0; }
        return *this;
    }
    T *data() { return p; }
    const T *data() const { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    void clear() { n = 0; }
    void reserve(size_t want)
    {
        if (want <= cap) return;
        const size_t kHuge = size_t(2) << 20;
        const size_t old_bytes = cap * sizeof(T);
        const size_t new_bytes = (want * sizeof(T) + kHuge - 1) / kHuge * kHuge;
        void *q = p ? mremap(p, old_bytes, new_bytes, MREMAP_MAYMOVE)
                    : mmap(nullptr, new_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (q == MAP_FAILED) throw std::bad_alloc();
        (void)madvise(q, new_bytes, MADV_HUGEPAGE);
        p = (T *)q;
        cap = new_bytes / sizeof(T);
    }
    // size becomes m; elements beyond the old size are unspecified (fresh pages read as zero)
    void resize_uninit(size_t m)
    {
        if (m > cap) reserve(m > 2 * cap ? m : 2 * cap);
        n = m;
    }
    void append(const T *src, size_t k)
    {
        const size_t at = n;
        resize_uninit(n + k);
        if (k) memcpy(p + at, src, k * sizeof(T));
    }
    void swap(HugeVec &o) noexcept { std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap); }
private:
    void release() { if (p) munmap(p, cap * sizeof(T)); p = nullptr; n = cap = 0; }
    T *p = nullptr;
    size_t n = 0, cap = 0;
};
