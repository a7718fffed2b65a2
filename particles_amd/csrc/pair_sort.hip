// pair_sort.hip — the one library primitive of the build: rocPRIM's stable
// least-significant-digit radix sort, used to order particles by cut-off cell
// for the binned pair term (reorder_kernels.hpp, cell_key_kernel).  Kept in its
// own translation unit so that the rocPRIM templates are compiled once.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdint.h>

namespace pcl {

hipError_t radix_sort_pairs_u32(void* temp, size_t* temp_bytes, const uint32_t* keys_in, uint32_t* keys_out,
                                const uint32_t* values_in, uint32_t* values_out, size_t n, unsigned end_bit,
                                hipStream_t stream)
{
    return rocprim::radix_sort_pairs(temp, *temp_bytes, keys_in, keys_out, values_in, values_out, n, 0u, end_bit,
                                     stream);
}

}  // namespace pcl
