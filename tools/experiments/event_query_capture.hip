// What does hipEventQuery say about an event that was recorded EAGERLY on a stream which is capturing NOW?
// (Root cause probe for the process-group watchdog abort "operation not permitted on an event last recorded in a capturing
// stream": the watchdog polls the end events of earlier, eager collectives while a later step is being captured.)
//   hipcc --offload-arch=gfx950 -O2 tools/experiments/event_query_capture.hip -o tools/experiments/event_query_capture
#include <hip/hip_runtime.h>
#include <cstdio>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k(int* p) { atomicAdd(p, 1); }
static const char* q(hipEvent_t e) { hipError_t r = hipEventQuery(e); (void)hipGetLastError(); return hipGetErrorName(r); }
int main() {
    int* d; CK(hipMalloc(&d, 4));
    hipStream_t cap, comm; CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&comm, hipStreamNonBlocking));
    hipEvent_t eager_on_comm, eager_on_cap, join, fork_back;
    CK(hipEventCreateWithFlags(&eager_on_comm, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&eager_on_cap, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&fork_back, hipEventDisableTiming));
    hipLaunchKernelGGL(k, 1, 1, 0, comm, d); CK(hipEventRecord(eager_on_comm, comm));
    hipLaunchKernelGGL(k, 1, 1, 0, cap, d); CK(hipEventRecord(eager_on_cap, cap));
    CK(hipDeviceSynchronize());
    printf("before capture        : eager_on_comm %s, eager_on_cap %s\n", q(eager_on_comm), q(eager_on_cap));
    CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(k, 1, 1, 0, cap, d);
    printf("capturing (cap only)  : eager_on_comm %s, eager_on_cap %s\n", q(eager_on_comm), q(eager_on_cap));
    CK(hipEventRecord(join, cap)); CK(hipStreamWaitEvent(comm, join, 0));           // comm joins the capture (what a captured collective does)
    hipLaunchKernelGGL(k, 1, 1, 0, comm, d);
    printf("capturing (comm joined): eager_on_comm %s, eager_on_cap %s   [same thread]\n", q(eager_on_comm), q(eager_on_cap));
    std::thread([&] { printf("capturing (comm joined): eager_on_comm %s, eager_on_cap %s   [other thread = the watchdog's view]\n", q(eager_on_comm), q(eager_on_cap)); }).join();
    CK(hipEventRecord(fork_back, comm)); CK(hipStreamWaitEvent(cap, fork_back, 0));
    hipGraph_t g; CK(hipStreamEndCapture(cap, &g));
    printf("after capture         : eager_on_comm %s, eager_on_cap %s, captured join %s\n", q(eager_on_comm), q(eager_on_cap), q(join));
    return 0;
}
