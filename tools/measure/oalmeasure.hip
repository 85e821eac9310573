// tools/measure/liboalmeasure.so -- MEASUREMENT AIDS written against the PUBLIC C-ABI of liboalgpu.so only (oalgpu_measure.h).
// Not part of the product: bench.py and tools/ load it to state what the boundary costs without a language binding in the loop.
#include "oalgpu_measure.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <vector>

namespace {
__global__ void EmptyKernel() { }
}

extern "C" {

__attribute__((visibility("default")))
int oalmeasure_pipelined_run(oalgpu_context *c, const oalgpu_voice_move *moves, size_t count, uint32_t move_sets,
    uint32_t updates, uint32_t samples_to_do, int post_process, float *out, size_t out_floats, double *wall_s, double *busy_s)
{
    if(!c || !moves || !out || count == 0 || move_sets == 0 || updates < 3) return OALGPU_ERR_INVALID;
    if(int rc = oalgpu_sync(c)) return rc;
    using clk = std::chrono::steady_clock;
    std::vector<uint32_t> tickets(updates);
    double waited = 0.0;
    const auto t0 = clk::now();
    for(uint32_t u = 0; u < updates; ++u)
    {
        if(int rc = oalgpu_voice_move_async(c, moves + size_t{u % move_sets} * count, count)) return rc;
        if(int rc = oalgpu_mix_update(c, samples_to_do, post_process)) return rc;
        if(int rc = oalgpu_read_output_async(c, &tickets[u])) return rc;
        if(u >= 2)
        {
            const auto w0 = clk::now();
            if(int rc = oalgpu_output_wait(c, tickets[u - 2], out, out_floats)) return rc;
            waited += std::chrono::duration<double>(clk::now() - w0).count();
        }
    }
    for(uint32_t u = updates - 2; u < updates; ++u)
    {
        const auto w0 = clk::now();
        if(int rc = oalgpu_output_wait(c, tickets[u], out, out_floats)) return rc;
        waited += std::chrono::duration<double>(clk::now() - w0).count();
    }
    const double wall = std::chrono::duration<double>(clk::now() - t0).count();
    if(wall_s) *wall_s = wall;
    if(busy_s) *busy_s = wall - waited;
    return OALGPU_OK;
}

__attribute__((visibility("default")))
int oalmeasure_submit_cost(oalgpu_context *c, const oalgpu_voice_move *moves, size_t count, uint32_t move_sets,
    uint32_t updates, uint32_t samples_to_do, int post_process, float *out, size_t out_floats, double *submit_s)
{
    if(!c || !moves || !out || !submit_s || count == 0 || move_sets == 0 || updates == 0) return OALGPU_ERR_INVALID;
    if(int rc = oalgpu_sync(c)) return rc;
    using clk = std::chrono::steady_clock;
    double spent = 0.0;
    for(uint32_t u = 0; u < updates; ++u)
    {
        uint32_t ticket = 0;
        const auto t0 = clk::now();
        if(int rc = oalgpu_voice_move_async(c, moves + size_t{u % move_sets} * count, count)) return rc;
        if(int rc = oalgpu_mix_update(c, samples_to_do, post_process)) return rc;
        if(int rc = oalgpu_read_output_async(c, &ticket)) return rc;
        spent += std::chrono::duration<double>(clk::now() - t0).count();
        if(int rc = oalgpu_output_wait(c, ticket, out, out_floats)) return rc;
    }
    *submit_s = spent / updates;
    return OALGPU_OK;
}

__attribute__((visibility("default")))
int oalmeasure_event_floor_ms(oalgpu_context *c, uint32_t reps, float *ms)
{
    if(!c || !ms || reps == 0 || reps > 4096) return OALGPU_ERR_INVALID;
    if(int rc = oalgpu_sync(c)) return rc;
    void *bus = nullptr, *stream = nullptr;
    size_t nfloats = 0;
    if(int rc = oalgpu_bus_device_ptr(c, &bus, &nfloats, &stream)) return rc;      // (the context's own stream)
    hipEvent_t a = nullptr, b = nullptr;
    if(hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return OALGPU_ERR_HIP;
    std::vector<float> each(reps);
    int rc = OALGPU_OK;
    for(uint32_t r = 0; r < reps && rc == OALGPU_OK; ++r)
    {
        hipExtLaunchKernelGGL(EmptyKernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), a, b, 0u);
        if(hipGetLastError() != hipSuccess || hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&each[r], a, b) != hipSuccess) rc = OALGPU_ERR_HIP;
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    if(rc) return rc;
    std::sort(each.begin(), each.end());
    *ms = each[reps / 2];
    return OALGPU_OK;
}

} // extern "C"
