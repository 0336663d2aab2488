// host_copy_probe.hip -- dev probe: how fast can 32 B/ray go to the device and 16 B/ray come back from
// PAGEABLE host memory?  (a) one hipMemcpy each way (what vt_trace_closest did), (b) chunked on two streams,
// (c) chunked through pinned staging filled/drained by OpenMP memcpy.
// Build: hipcc -O2 -fopenmp --offload-arch=gfx950 scripts/host_copy_probe.hip -o scripts/_build/host_copy_probe
#include <hip/hip_runtime.h>
#include <omp.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void par_copy(void* dst, const void* src, size_t bytes, int threads)
{
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long long o = 0; o < (long long)bytes; o += (1 << 20)) {
        const size_t m = bytes - o < (1u << 20) ? bytes - o : (1u << 20);
        std::memcpy((char*)dst + o, (const char*)src + o, m);
    }
}

int main(int argc, char** argv)
{
    const size_t n = size_t(1) << 24;
    const size_t in_b = n * 32, out_b = n * 16;
    char* in = (char*)std::malloc(in_b);
    char* out = (char*)std::malloc(out_b);
    std::memset(in, 1, in_b); std::memset(out, 0, out_b);
    char *d_in, *d_out;
    hipMalloc(&d_in, in_b); hipMalloc(&d_out, out_b);
    hipMemset(d_out, 2, out_b);
    hipStream_t s0, s1;
    hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);

    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        hipMemcpyAsync(d_in, in, in_b, hipMemcpyHostToDevice, s0);
        hipMemcpyAsync(out, d_out, out_b, hipMemcpyDeviceToHost, s0);
        hipStreamSynchronize(s0);
        double t1 = now();
        if (rep) printf("(a) one copy each way, one stream        : %7.2f ms  -> %6.0f Mrays/s\n", (t1 - t0) * 1e3, n / (t1 - t0) / 1e6);
    }
    for (size_t chunk : {size_t(1) << 20, size_t(1) << 22}) {
        double t0 = now();
        for (size_t off = 0; off < n; off += chunk) {
            hipMemcpyAsync(d_in + off * 32, in + off * 32, chunk * 32, hipMemcpyHostToDevice, s0);
            hipMemcpyAsync(out + off * 16, d_out + off * 16, chunk * 16, hipMemcpyDeviceToHost, s1);
        }
        hipStreamSynchronize(s0); hipStreamSynchronize(s1);
        double t1 = now();
        printf("(b) pageable, %zu Mi-ray chunks, two streams : %7.2f ms  -> %6.0f Mrays/s\n", chunk >> 20, (t1 - t0) * 1e3, n / (t1 - t0) / 1e6);
    }
    for (int threads : {4, 8, 16, 32}) {
        if (threads > omp_get_max_threads()) break;
        const size_t chunk = size_t(1) << 20;
        char *pin_in[2], *pin_out[2];
        for (int k = 0; k < 2; ++k) { hipHostMalloc((void**)&pin_in[k], chunk * 32); hipHostMalloc((void**)&pin_out[k], chunk * 16); }
        hipEvent_t in_done[2], out_done[2];
        for (int k = 0; k < 2; ++k) { hipEventCreate(&in_done[k]); hipEventCreate(&out_done[k]); }
        for (int rep = 0; rep < 2; ++rep) {
            double t0 = now();
            const size_t nchunks = n / chunk;
            for (size_t c = 0; c <= nchunks; ++c) {
                const int b = int(c & 1);
                if (c < nchunks) {
                    if (c >= 2) hipEventSynchronize(in_done[b]);
                    par_copy(pin_in[b], in + c * chunk * 32, chunk * 32, threads);
                    hipMemcpyAsync(d_in + c * chunk * 32, pin_in[b], chunk * 32, hipMemcpyHostToDevice, s0);
                    hipEventRecord(in_done[b], s0);
                }
                if (c >= 1) {   // drain the previous chunk's results
                    const int pb = int((c - 1) & 1);
                    hipEventSynchronize(out_done[pb]);
                    par_copy(out + (c - 1) * chunk * 16, pin_out[pb], chunk * 16, threads);
                }
                if (c < nchunks) {
                    hipStreamWaitEvent(s1, in_done[b], 0);
                    hipMemcpyAsync(pin_out[b], d_out + c * chunk * 16, chunk * 16, hipMemcpyDeviceToHost, s1);
                    hipEventRecord(out_done[b], s1);
                }
            }
            double t1 = now();
            if (rep) printf("(c) pinned staging, 1 Mi-ray chunks, %2d copy threads: %7.2f ms  -> %6.0f Mrays/s\n", threads, (t1 - t0) * 1e3, n / (t1 - t0) / 1e6);
        }
        for (int k = 0; k < 2; ++k) { hipHostFree(pin_in[k]); hipHostFree(pin_out[k]); }
    }
    printf("check %d\n", out[out_b - 1]);
    return 0;
}
