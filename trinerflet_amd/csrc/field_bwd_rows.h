// field_bwd_rows.h -- internal (not part of the C ABI): launch of k_field_bwd_rows (field_bwd_rows.hip) from field_bwd.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// hidden 64, C = 16 or 32, binned mode: one fp32 weight-gradient slab per workgroup into `workspace`, *nslab of them;
// the caller sums them (k_slab_reduce)
int tnl_bwd_rows_launch(int C, const float* gsig, const float* grgb, const void* feats, const float* dirs, uint32_t M,
                        const void* packed, void* workspace, const int32_t* m_actual, void* dfeat, hipStream_t st,
                        uint32_t* nslab);
