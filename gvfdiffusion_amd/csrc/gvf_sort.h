// gvf_sort.h -- internal declarations of the radix sort used by the rasteriser.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

int gvf_sort_num_blocks(int64_t n_cap);
extern "C" size_t gvf_sort_tmp_bytes(int64_t n);
// Sorts pairs on key bits [begin_bit,end_bit); element count min(*n_ptr, n_cap) is read on the device.
// *result_in_alt = 1 when the sorted data ended in (keys_alt, vals_alt).
int gvf_sort_pairs_device_n(uint64_t* keys, uint64_t* keys_alt, uint32_t* vals, uint32_t* vals_alt,
                            const uint32_t* n_ptr, int64_t n_cap, int begin_bit, int end_bit, void* tmp,
                            size_t tmp_bytes, hipStream_t stream, int* result_in_alt);
