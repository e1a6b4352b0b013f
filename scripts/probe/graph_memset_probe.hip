// Probe for DESIGN.md 4c "Clears are kernels, not memsets": what does a stream capture make of hipMemsetAsync, and does a
// replayed graph clear the buffer before the kernel that depends on it -- on EVERY replay?
//
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/graph_memset_probe scripts/probe/graph_memset_probe.hip && /tmp/graph_memset_probe
//
// For each case: capture { memset(s) ; kernel } on one stream, print the graph's nodes (type; for memset nodes dst,
// elementSize, width, height, pitch, value) and edges, then replay 4 times.  The kernel counts the words of the range that
// are NOT zero when it runs (what a launch that trusts the clear would trip over) and then fills the range with a pattern,
// so that a replay whose memset does not run -- or runs after the kernel -- is seen from the second replay on; the first
// one starts from a zero allocation, as the encode's workspace did.
// Cases mirror what round 2's failing encode issued (commit 87e5955^): four memsets in a row of B*4, B*33*8, B*NBLK*8,
// B*cells*8 bytes, a 4-byte memset between kernels, and one of sizeof(ctl) + 32 KiB + queue bytes at an address 256-byte
// aligned inside a larger allocation; plus a forked capture (the memset on the origin stream, the kernel on a stream
// that waits on an event recorded after it) as the sub-batch form had.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__device__ unsigned long long g_sample;   // the first non-zero 8 bytes any kernel found (what the range was filled with)
__global__ void count_then_dirty(unsigned *p, size_t words, unsigned *nonzero, unsigned pattern) {
    unsigned bad = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) {
        if (p[i] != 0u) {
            ++bad;
            if (i + 1 < words && (i & 1) == 0) atomicCAS(&g_sample, 0ull, (unsigned long long)p[i] | ((unsigned long long)p[i + 1] << 32));
        }
        p[i] = pattern;
    }
    if (bad) atomicAdd(nonzero, bad);
}

__global__ void touch(unsigned *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 0u; }

static const char *type_name(hipGraphNodeType t) {
    switch (t) {
        case hipGraphNodeTypeKernel: return "kernel";
        case hipGraphNodeTypeMemcpy: return "memcpy";
        case hipGraphNodeTypeMemset: return "memset";
        case hipGraphNodeTypeHost: return "host";
        case hipGraphNodeTypeGraph: return "graph";
        case hipGraphNodeTypeEmpty: return "empty";
        case hipGraphNodeTypeWaitEvent: return "wait_event";
        case hipGraphNodeTypeEventRecord: return "event_record";
        default: return "other";
    }
}

static void dump(hipGraph_t g) {
    size_t n = 0;
    CHECK(hipGraphGetNodes(g, nullptr, &n));
    std::vector<hipGraphNode_t> nodes(n);
    CHECK(hipGraphGetNodes(g, nodes.data(), &n));
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType t;
        CHECK(hipGraphNodeGetType(nodes[i], &t));
        printf("    node %zu: %s", i, type_name(t));
        if (t == hipGraphNodeTypeMemset) {
            hipMemsetParams mp;
            CHECK(hipGraphMemsetNodeGetParams(nodes[i], &mp));
            printf("  dst=%p elementSize=%u width=%zu height=%zu pitch=%zu value=%u", mp.dst, mp.elementSize, mp.width, mp.height, mp.pitch, mp.value);
        }
        size_t nd = 0;
        CHECK(hipGraphNodeGetDependencies(nodes[i], nullptr, &nd));
        std::vector<hipGraphNode_t> deps(nd);
        if (nd) CHECK(hipGraphNodeGetDependencies(nodes[i], deps.data(), &nd));
        printf("  deps:");
        for (size_t d = 0; d < nd; ++d)
            for (size_t j = 0; j < n; ++j) if (nodes[j] == deps[d]) printf(" %zu", j);
        printf("\n");
    }
}

struct Range { size_t offset, bytes; };

// one case: memsets over `ranges` of one allocation, then the counting kernel over each range
static int run_case(const char *name, const std::vector<Range> &ranges, bool forked, bool kernel_between, int churn = 0, int extra_nodes = 0, int copies = 0) {
    printf("case %s\n", name);
    size_t total = 0;
    for (auto &r : ranges) total = std::max(total, r.offset + r.bytes);
    total += 4096;
    char *base = nullptr;
    unsigned *nonzero = nullptr;
    CHECK(hipMalloc(&base, total));
    CHECK(hipMalloc(&nonzero, 4 * ranges.size()));
    CHECK(hipMemset(base, 0, total));
    CHECK(hipMemset(nonzero, 0, 4 * ranges.size()));
    hipStream_t s, s2;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t ev, ev2;
    CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    CHECK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
    CHECK(hipDeviceSynchronize());
    hipGraph_t graph;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (size_t i = 0; i < ranges.size(); ++i) {
        CHECK(hipMemsetAsync(base + ranges[i].offset, 0, ranges[i].bytes, s));
        if (kernel_between) hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s, nonzero);
    }
    hipStream_t ks = s;
    if (forked) {
        CHECK(hipEventRecord(ev, s));
        CHECK(hipStreamWaitEvent(s2, ev, 0));
        ks = s2;
    }
    for (size_t i = 0; i < ranges.size(); ++i)
        hipLaunchKernelGGL(count_then_dirty, dim3(64), dim3(256), 0, ks, reinterpret_cast<unsigned *>(base + ranges[i].offset),
                           ranges[i].bytes / 4, nonzero + i, 0xdeadbeefu);
    for (int i = 0; i < extra_nodes; ++i) hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, ks, nonzero);   // a long graph, as an encode is
    if (forked) {
        CHECK(hipEventRecord(ev2, s2));
        CHECK(hipStreamWaitEvent(s, ev2, 0));
    }
    CHECK(hipStreamEndCapture(s, &graph));
    if (extra_nodes) printf("    (+ %d kernel nodes behind them)\n", extra_nodes); else dump(graph);
    hipGraphExec_t exec;
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    int failures = 0;
    unsigned *scratch = nullptr;
    void *pinned = nullptr;
    CHECK(hipHostMalloc(&pinned, 4096, 0));
    CHECK(hipMalloc(&scratch, 256));
    CHECK(hipMemset(scratch, 0, 256));
    for (int rep = 0; rep < 4; ++rep) {
        // what a host program does between two replays: other launches on the same stream (their kernel arguments -- here a
        // device pointer -- go through the stream's argument buffers)
        for (int i = 0; i < churn; ++i) hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s, scratch);
        // ... and reads results back: small asynchronous device-to-pinned-host copies (what tensor.item() / torch.equal do)
        for (int i = 0; i < copies; ++i) CHECK(hipMemcpyAsync(pinned, scratch + (i & 15), 8, hipMemcpyDeviceToHost, s));
        if (copies) CHECK(hipStreamSynchronize(s));
        CHECK(hipMemsetAsync(nonzero, 0, 4 * ranges.size(), s));
        CHECK(hipGraphLaunch(exec, s));
        CHECK(hipStreamSynchronize(s));
        std::vector<unsigned> h(ranges.size());
        CHECK(hipMemcpy(h.data(), nonzero, 4 * ranges.size(), hipMemcpyDeviceToHost));
        printf("    replay %d: words found non-zero by the kernel, per range:", rep);
        for (size_t i = 0; i < ranges.size(); ++i) { printf(" %u/%zu", h[i], ranges[i].bytes / 4); failures += h[i] != 0; }
        printf("\n");
    }
    void *base_keep = base, *scratch_keep = scratch;
    CHECK(hipFree(scratch));
    CHECK(hipGraphExecDestroy(exec));
    CHECK(hipGraphDestroy(graph));
    CHECK(hipFree(base));
    CHECK(hipFree(nonzero));
    unsigned long long sample = 0, zero = 0;
    CHECK(hipMemcpyFromSymbol(&sample, HIP_SYMBOL(g_sample), 8));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_sample), &zero, 8));
    if (failures) printf("    first non-zero 8 bytes found: 0x%016llx (the kernel's own pattern would be 0xdeadbeefdeadbeef; base %p, device source of the copies %p, their pinned destination %p)\n", sample, (void *)base_keep, (void *)scratch_keep, pinned);
    CHECK(hipHostFree(pinned));
    printf("    -> %s\n", failures ? "STALE DATA SEEN" : "ok");
    return failures;
}

int main() {
    int dev = 0;
    CHECK(hipSetDevice(dev));
    int rt = 0;
    CHECK(hipRuntimeGetVersion(&rt));
    printf("hip runtime %d\n", rt);
    const size_t B = 50, NBLK = 94, NAT = 2, K = 9;
    int bad = 0;
    // the four clears of round 2's fft_setup at the graph test's shape (64 x 256 dictionary, 50 segments of 6000 samples)
    std::vector<Range> setup = {{0, B * 4}, {4096, B * 33 * 8}, {4096 + 16384, B * NBLK * 2 * 4}, {4096 + 16384 + 65536, B * NBLK * NAT * 8}};
    bad += run_case("fft_setup: four memsets in a row, then kernels", setup, false, false);
    bad += run_case("the same with a kernel between the memsets", setup, false, true);
    // the persistent launch's control block + ticket lines + queue, 256-byte aligned inside a larger block
    const size_t pctl = 256 + 512 * 64 + (B * (K - 1) + 2) * 128;
    bad += run_case("persistent control block (one memset, 84 KB)", {{256, pctl}}, false, false);
    bad += run_case("persistent control block at an odd offset (4-byte aligned)", {{260, pctl}}, false, false);
    bad += run_case("4-byte memset (dscale)", {{512, 4}}, false, true);
    bad += run_case("forked: memset on the origin stream, kernels on a stream that waits for it", setup, true, false);
    bad += run_case("large memset (64 MiB)", {{0, (size_t)64 << 20}}, false, false);
    bad += run_case("headline shape keys (64 x 512 x 16 x 8 B) + control block", {{0, (size_t)64 * 512 * 16 * 8}, {(size_t)8 << 20, 256 + 512 * 64 + (64 * 63 + 2) * 128}}, false, false);
    // what the captured encode had and the cases above lack: other launches between the replays, and a long graph
    const std::vector<Range> ctl = {{256, pctl}};
    bad += run_case("control block, 64 other launches on the stream between replays", ctl, false, false, 64, 0);
    bad += run_case("control block, 4096 other launches between replays", ctl, false, false, 4096, 0);
    bad += run_case("control block, 20000 other launches between replays", ctl, false, false, 20000, 0);
    bad += run_case("control block in a graph of 400 kernel nodes", ctl, false, false, 0, 400);
    bad += run_case("control block in a graph of 400 kernel nodes, 4096 other launches between replays", ctl, false, false, 4096, 400);
    bad += run_case("four memsets, kernel between, 400 nodes, 4096 launches between replays", setup, false, true, 4096, 400);
    // ... and what a host program that LOOKS at its results does between replays
    bad += run_case("control block, 1 small device-to-pinned-host copy between replays", ctl, false, false, 0, 0, 1);
    bad += run_case("control block, 64 such copies between replays", ctl, false, false, 0, 0, 64);
    bad += run_case("four memsets, kernel between, 64 copies between replays", setup, false, true, 0, 0, 64);
    printf("%s\n", bad ? "RESULT: stale data after a replayed memset node" : "RESULT: every replay saw cleared memory");
    return 0;
}
