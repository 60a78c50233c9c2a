// rocPRIM-backed device sort / scan used by the reorder, the cell list and the border lists.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace meso {
size_t sort_temp_bytes_u32(int n);
size_t sort_temp_bytes_u64(int n);
size_t scan_temp_bytes(int n);
// key/value double buffers are swapped in place: on return keys/vals point at the sorted data
hipError_t sort_pairs_u32(void *temp, size_t temp_bytes, uint32_t *&keys, uint32_t *&keys_alt, int *&vals,
                          int *&vals_alt, int n, int bits, hipStream_t s);
hipError_t sort_pairs_u64(void *temp, size_t temp_bytes, uint64_t *&keys, uint64_t *&keys_alt, int *&vals,
                          int *&vals_alt, int n, int bits, hipStream_t s);
hipError_t exclusive_scan_i32(void *temp, size_t temp_bytes, const int *in, int *out, int n, hipStream_t s);
} // namespace meso
