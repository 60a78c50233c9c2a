// rocPRIM radix sort / scan wrappers (replace SortPlan<16,...>, /root/reference/src/USER-MESO/sort_meso.h:350-429,
// whose 4-bit LSD passes assert warp==32 at :49).  Stream-ordered, caller-owned temp storage.
#include <cstring>
#include <string.h>
#include "sort.h"
#include <rocprim/rocprim.hpp>

namespace meso {

// rocPRIM routes up to 1 048 576 items through its merge sort: one block sort + two launches per merge level.  Sorting
// 4096 items per block (default 1024) removes two levels = four of the ~6 us launches the rebuild is bound by.
using SortConfig = rocprim::radix_sort_config<rocprim::kernel_config<512, 8>, rocprim::merge_sort_config<512, 512, 8>,
                                              rocprim::default_config>;

size_t sort_temp_bytes_u32(int n)
{
    size_t bytes = 0;
    rocprim::double_buffer<uint32_t> k(nullptr, nullptr);
    rocprim::double_buffer<int> v(nullptr, nullptr);
    (void)rocprim::radix_sort_pairs<SortConfig>(nullptr, bytes, k, v, (size_t)n, 0, 32, (hipStream_t)0);
    return bytes;
}

size_t sort_temp_bytes_u64(int n)
{
    size_t bytes = 0;
    rocprim::double_buffer<unsigned long long> k(nullptr, nullptr);
    rocprim::double_buffer<int> v(nullptr, nullptr);
    (void)rocprim::radix_sort_pairs<SortConfig>(nullptr, bytes, k, v, (size_t)n, 0, 64, (hipStream_t)0);
    return bytes;
}

size_t scan_temp_bytes(int n)
{
    size_t bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, bytes, (const int *)nullptr, (int *)nullptr, 0, (size_t)n,
                                  rocprim::plus<int>(), (hipStream_t)0);
    return bytes;
}

hipError_t sort_pairs_u32(void *temp, size_t temp_bytes, uint32_t *&keys, uint32_t *&keys_alt, int *&vals,
                          int *&vals_alt, int n, int bits, hipStream_t s)
{
    rocprim::double_buffer<uint32_t> k(keys, keys_alt);
    rocprim::double_buffer<int> v(vals, vals_alt);
    hipError_t e = rocprim::radix_sort_pairs<SortConfig>(temp, temp_bytes, k, v, (size_t)n, 0, (unsigned)bits, s);
    keys = k.current(); keys_alt = k.alternate();
    vals = v.current(); vals_alt = v.alternate();
    return e;
}

hipError_t sort_pairs_u64(void *temp, size_t temp_bytes, uint64_t *&keys, uint64_t *&keys_alt, int *&vals,
                          int *&vals_alt, int n, int bits, hipStream_t s)
{
    rocprim::double_buffer<unsigned long long> k((unsigned long long *)keys, (unsigned long long *)keys_alt);
    rocprim::double_buffer<int> v(vals, vals_alt);
    hipError_t e = rocprim::radix_sort_pairs<SortConfig>(temp, temp_bytes, k, v, (size_t)n, 0, (unsigned)bits, s);
    keys = (uint64_t *)k.current(); keys_alt = (uint64_t *)k.alternate();
    vals = v.current(); vals_alt = v.alternate();
    return e;
}

hipError_t exclusive_scan_i32(void *temp, size_t temp_bytes, const int *in, int *out, int n, hipStream_t s)
{
    return rocprim::exclusive_scan(temp, temp_bytes, in, out, 0, (size_t)n, rocprim::plus<int>(), s);
}

} // namespace meso
