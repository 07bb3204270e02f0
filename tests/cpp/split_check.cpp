// Test helper: reads float32 values from stdin (binary), writes for each value
//   u16 f16(a), u16 bf16(a), u16 x 2 fp16x2 pieces, u16 x 3 bf16x3 pieces, f32 joined fp16x2, f32 joined bf16x3
// using the host routines the library uses to split weights (speechseparation_amd/csrc/split_host.h).
#include "split_host.h"
#include <cstdio>
#include <vector>
int main()
{
    std::vector<float> v;
    float x;
    while (fread(&x, 4, 1, stdin) == 1) v.push_back(x);
    for (float a : v) {
        uint16_t out[7];
        out[0] = bsrnn::f16_from_float(a);
        out[1] = bsrnn::bf16_from_float(a);
        bsrnn::split_planes_host(&a, 1, 2, out + 2);
        bsrnn::split_planes_host(&a, 1, 3, out + 4);
        float j[2] = {bsrnn::join_planes_host(out + 2, 1, 0, 2), bsrnn::join_planes_host(out + 4, 1, 0, 3)};
        fwrite(out, 2, 7, stdout);
        fwrite(j, 4, 2, stdout);
    }
    return 0;
}
