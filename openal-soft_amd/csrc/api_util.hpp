// Small host-side helpers shared by the C-ABI translation units (api.hip, conv_api.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#include "../../include/oalgpu.h"

namespace oalgpu {

// records the message for oalgpu_last_error() on this thread and returns `code`
int Fail(int code, const std::string &msg);
int UseDevice(int device);
// lines an effect instance mixes into (reverb_api.hip / conv_api.hip)
uint32_t ReverbOutLines(const oalgpu_reverb *r);
uint32_t ConvOutLines(const oalgpu_convolution *c);
uint32_t EffectOutLines(const oalgpu_effect *e);
uint32_t EffectInChannels(const oalgpu_effect *e);
} // namespace oalgpu
// the effect's block on device memory, asynchronous on `hip_stream` (effects_api.hip)
int oalgpu_effect_process_device(oalgpu_effect *e, void *hip_stream, const float *wet_in_dev, float *out_lines_dev, uint32_t n);
namespace oalgpu {

#define HIP_TRY(expr) do { \
    const hipError_t err_ = (expr); \
    if(err_ != hipSuccess) \
        return ::oalgpu::Fail(OALGPU_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(err_)); \
} while(0)

// device scratch that frees itself
template<typename T>
struct DevBuf {
    T *p{nullptr};
    size_t n{0};
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf &operator=(const DevBuf&) = delete;
    ~DevBuf() { if(p) (void)hipFree(p); }
    hipError_t alloc(size_t count)
    {
        if(p) { (void)hipFree(p); p = nullptr; }
        n = count;
        return hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T));
    }
    hipError_t upload(const T *src, size_t count) { return hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice); }
    hipError_t download(T *dst, size_t count) const { return hipMemcpy(dst, p, count * sizeof(T), hipMemcpyDeviceToHost); }
    hipError_t zero() { return hipMemset(p, 0, std::max<size_t>(n, 1) * sizeof(T)); }
};

} // namespace oalgpu
