// gpu_transport.h -- the HIP-runtime and RCCL calls the multi-GPU driver makes from the host (streams, events, device buffers,
// grouped ncclSend / ncclRecv), behind plain C++ signatures: the host's own float3/float4 (vec.h, standing in for CUDA's
// vector_types.h as the reference's host code uses them) and HIP's vector types cannot meet in one translation unit.
// Every function throws std::runtime_error on failure, like CUDA_CHECK does in the reference (sutil/Exception.h:93-157).
#pragma once
#include <cstddef>
#include <vector>

namespace engine {
namespace host {
namespace gpu {

void SetDevice(int device);
void DeviceSync();
void* StreamCreate(bool highestPriority);   // non-blocking stream on the current device
void StreamDestroy(void* stream);
void StreamSync(void* stream);
void* EventCreate();                        // timing disabled
void EventDestroy(void* event);
void EventRecord(void* event, void* stream);
void StreamWaitEvent(void* stream, void* event);
void* Malloc(size_t bytes);                 // zero-filled
void Free(void* p);
void CopyDeviceToDeviceAsync(void* dst, const void* src, size_t bytes, void* stream);
void CopyDeviceToHost(void* dst, const void* src, size_t bytes);

// RCCL: one process, one rank per listed GPU (ncclCommInitAll); rank i is devices[i]
struct Comms;
Comms* CommsCreate(const std::vector<int>& devices);
void CommsDestroy(Comms* comms);
void GroupStart();
void GroupEnd();
void Send(Comms* comms, int rank, int toRank, const void* buf, size_t bytes, void* stream);    // on rank's GPU and stream
void Recv(Comms* comms, int rank, int fromRank, void* buf, size_t bytes, void* stream);

}  // namespace gpu
}  // namespace host
}  // namespace engine
