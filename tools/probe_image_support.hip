// Does this device expose texture/image hardware to HIP?  (hard part 2: probe, do not assume.)
#include <hip/hip_runtime.h>
#include <cstdio>
int main()
{
    int v = -1, tex1d = -1, tex2dw = -1;
    hipError_t e = hipDeviceGetAttribute(&v, hipDeviceAttributeImageSupport, 0);
    hipDeviceGetAttribute(&tex1d, hipDeviceAttributeMaxTexture1DWidth, 0);
    hipDeviceGetAttribute(&tex2dw, hipDeviceAttributeMaxTexture2DWidth, 0);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s %s: hipDeviceAttributeImageSupport=%d (%s) maxTexture1D=%d maxTexture2DWidth=%d\n", p.name, p.gcnArchName, v,
           hipGetErrorString(e), tex1d, tex2dw);
    hipTextureObject_t tex = 0;
    hipResourceDesc rd{};
    float *d = nullptr;
    hipMalloc(&d, 1024 * sizeof(float4));
    rd.resType = hipResourceTypeLinear;
    rd.res.linear.devPtr = d;
    rd.res.linear.desc = hipCreateChannelDesc<float4>();
    rd.res.linear.sizeInBytes = 1024 * sizeof(float4);
    hipTextureDesc td{};
    td.readMode = hipReadModeElementType;
    e = hipCreateTextureObject(&tex, &rd, &td, nullptr);
    printf("hipCreateTextureObject(linear float4): %s\n", hipGetErrorString(e));
    return 0;
}
