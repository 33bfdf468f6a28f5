// field_h128.hip -- the hidden-128 (C = 48) instantiation of the fused field forward, as its own object: field.hip compiled
// with TNL_FIELD_H128_ONLY and -fno-slp-vectorize (build.py PER_FILE; the reason is at tnl_field_forward_h128 in field.hip).
#define TNL_FIELD_H128_ONLY 1
#include "field.hip"
