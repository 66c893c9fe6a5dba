// gpu_transport.cpp -- see gpu_transport.h.  The only host source that includes HIP and RCCL headers.
#include "gpu_transport.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <stdexcept>
#include <string>

namespace engine {
namespace host {
namespace gpu {

namespace {
void hipCheck(hipError_t e, const char* what)
{
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + " failed: " + hipGetErrorString(e));
}
void ncclCheck(ncclResult_t r, const char* what)
{
    if (r != ncclSuccess) throw std::runtime_error(std::string(what) + " failed: " + ncclGetErrorString(r));
}
}  // namespace

struct Comms {
    std::vector<int> devices;
    std::vector<ncclComm_t> comm;
};

void SetDevice(int device) { hipCheck(hipSetDevice(device), "hipSetDevice"); }
void DeviceSync() { hipCheck(hipDeviceSynchronize(), "hipDeviceSynchronize"); }

void* StreamCreate(bool highestPriority)
{
    hipStream_t s = nullptr;
    if (highestPriority) {
        // RCCL's kernels should not queue behind the persistent megakernel
        int lo = 0, hi = 0;
        hipCheck(hipDeviceGetStreamPriorityRange(&lo, &hi), "hipDeviceGetStreamPriorityRange");
        hipCheck(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi), "hipStreamCreateWithPriority");
    } else {
        hipCheck(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreateWithFlags");
    }
    return s;
}
void StreamDestroy(void* stream) { (void)hipStreamDestroy(static_cast<hipStream_t>(stream)); }
void StreamSync(void* stream) { hipCheck(hipStreamSynchronize(static_cast<hipStream_t>(stream)), "hipStreamSynchronize"); }

void* EventCreate()
{
    hipEvent_t e = nullptr;
    hipCheck(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreateWithFlags");
    return e;
}
void EventDestroy(void* event) { (void)hipEventDestroy(static_cast<hipEvent_t>(event)); }
void EventRecord(void* event, void* stream) { hipCheck(hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(stream)), "hipEventRecord"); }
void StreamWaitEvent(void* stream, void* event)
{
    hipCheck(hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(event), 0), "hipStreamWaitEvent");
}

void* Malloc(size_t bytes)
{
    void* p = nullptr;
    hipCheck(hipMalloc(&p, bytes), "hipMalloc");
    hipCheck(hipMemset(p, 0, bytes), "hipMemset");
    // hipMemset of device memory is asynchronous on the null stream, and the streams this renderer works on do not wait for that
    // stream: without this the zeros could land AFTER the first kernel that writes the buffer (seen once in ~10 runs of the GPU suite:
    // ReadAccum allocates its gather buffers on first use and reads all zeros)
    hipCheck(hipStreamSynchronize(nullptr), "hipStreamSynchronize");
    return p;
}
void Free(void* p) { (void)hipFree(p); }
void CopyDeviceToDeviceAsync(void* dst, const void* src, size_t bytes, void* stream)
{
    hipCheck(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)), "hipMemcpyAsync");
}
void CopyDeviceToHost(void* dst, const void* src, size_t bytes) { hipCheck(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost), "hipMemcpy"); }

Comms* CommsCreate(const std::vector<int>& devices)
{
    Comms* c = new Comms();
    c->devices = devices;
    c->comm.assign(devices.size(), nullptr);
    const ncclResult_t r = ncclCommInitAll(c->comm.data(), static_cast<int>(devices.size()), devices.data());
    if (r != ncclSuccess) {
        delete c;
        ncclCheck(r, "ncclCommInitAll");
    }
    return c;
}
void CommsDestroy(Comms* c)
{
    if (!c) return;
    for (size_t i = 0; i < c->comm.size(); ++i)
        if (c->comm[i]) {
            (void)hipSetDevice(c->devices[i]);
            (void)ncclCommDestroy(c->comm[i]);
        }
    delete c;
}
void GroupStart() { ncclCheck(ncclGroupStart(), "ncclGroupStart"); }
void GroupEnd() { ncclCheck(ncclGroupEnd(), "ncclGroupEnd"); }
void Send(Comms* c, int rank, int toRank, const void* buf, size_t bytes, void* stream)
{
    ncclCheck(ncclSend(buf, bytes, ncclUint8, toRank, c->comm[static_cast<size_t>(rank)], static_cast<hipStream_t>(stream)), "ncclSend");
}
void Recv(Comms* c, int rank, int fromRank, void* buf, size_t bytes, void* stream)
{
    ncclCheck(ncclRecv(buf, bytes, ncclUint8, fromRank, c->comm[static_cast<size_t>(rank)], static_cast<hipStream_t>(stream)), "ncclRecv");
}

}  // namespace gpu
}  // namespace host
}  // namespace engine
