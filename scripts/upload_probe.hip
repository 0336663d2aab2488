// upload_probe.hip -- dev probe for the batch boundary (accel:TraverseBatch(buffer), 1 Mi rays = 32 MB in, 16 MB out): what do
// the ways of getting a pageable caller buffer to the device cost on THIS host?  (a) one pageable hipMemcpy, (b) hipHostRegister
// + DMA + hipHostUnregister, (c) staging through pinned buffers with T copy threads in 256 Ki-ray chunks, (d) a single-thread
// memcpy / std::string::assign of the result.
// Build: hipcc -O2 -fopenmp --offload-arch=gfx950 scripts/upload_probe.hip -o scripts/_build/upload_probe
#include <hip/hip_runtime.h>
#include <omp.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void par_copy(void* dst, const void* src, size_t bytes, int threads)
{
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long long o = 0; o < (long long)bytes; o += (1 << 20)) {
        const size_t m = bytes - o < (1u << 20) ? bytes - o : (1u << 20);
        std::memcpy((char*)dst + o, (const char*)src + o, m);
    }
}

int main()
{
    const size_t n = size_t(1) << 20, in_b = n * 32, out_b = n * 16;
    printf("hardware_concurrency %u, omp_get_max_threads %d\n", std::thread::hardware_concurrency(), omp_get_max_threads());
    char* in = (char*)std::malloc(in_b);
    std::memset(in, 1, in_b);
    char *d_in, *d_out, *pin, *pin_out;
    hipMalloc(&d_in, in_b); hipMalloc(&d_out, out_b); hipMemset(d_out, 3, out_b);
    hipHostMalloc((void**)&pin, in_b); hipHostMalloc((void**)&pin_out, out_b);
    hipStream_t s0; hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        hipMemcpy(d_in, in, in_b, hipMemcpyHostToDevice);
        double t1 = now();
        if (rep) printf("(a) pageable hipMemcpy H2D 32 MB           : %6.2f ms\n", (t1 - t0) * 1e3);
    }
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        hipError_t e = hipHostRegister(in, in_b, hipHostRegisterDefault);
        double t1 = now();
        hipMemcpyAsync(d_in, in, in_b, hipMemcpyHostToDevice, s0); hipStreamSynchronize(s0);
        double t2 = now();
        hipHostUnregister(in);
        double t3 = now();
        if (rep) printf("(b) hipHostRegister %6.2f ms (%s), DMA %6.2f ms, unregister %6.2f ms\n", (t1 - t0) * 1e3, hipGetErrorString(e), (t2 - t1) * 1e3, (t3 - t2) * 1e3);
    }
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        hipMemcpyAsync(d_in, pin, in_b, hipMemcpyHostToDevice, s0); hipStreamSynchronize(s0);
        double t1 = now();
        hipMemcpyAsync(pin_out, d_out, out_b, hipMemcpyDeviceToHost, s0); hipStreamSynchronize(s0);
        double t2 = now();
        if (rep) printf("    pinned H2D 32 MB %6.2f ms (%.1f GB/s), pinned D2H 16 MB %6.2f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, in_b / (t1 - t0) / 1e9, (t2 - t1) * 1e3, out_b / (t2 - t1) / 1e9);
    }
    for (int threads : {1, 2, 4, 8}) {
        if (threads > omp_get_max_threads()) break;
        for (size_t chunk : {size_t(1) << 18, size_t(1) << 20}) {
            for (int rep = 0; rep < 3; ++rep) {
                double t0 = now();
                for (size_t off = 0; off < n; off += chunk) {
                    par_copy(pin + off * 32, in + off * 32, chunk * 32, threads);
                    hipMemcpyAsync(d_in + off * 32, pin + off * 32, chunk * 32, hipMemcpyHostToDevice, s0);
                }
                hipStreamSynchronize(s0);
                double t1 = now();
                if (rep == 2) printf("(c) staged, %d copy threads, %4zu Ki-ray chunks : %6.2f ms\n", threads, chunk >> 10, (t1 - t0) * 1e3);
            }
        }
    }
    {   // registration of a large caller buffer on the fly, DMA straight from / to it on two streams (no staging copies at all)
        const size_t big_n = size_t(1) << 24, bin = big_n * 32, bout = big_n * 16;
        char* bi = (char*)std::malloc(bin); char* bo = (char*)std::malloc(bout);
        std::memset(bi, 1, bin);                                  // the rays were written by the caller; the result array is fresh
        char *dbi, *dbo; hipMalloc(&dbi, bin); hipMalloc(&dbo, bout);
        hipStream_t s1; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
        for (int rep = 0; rep < 3; ++rep) {
            double t0 = now();
            hipError_t e1 = hipHostRegister(bi, bin, hipHostRegisterDefault), e2 = hipHostRegister(bo, bout, hipHostRegisterDefault);
            double t1 = now();
            const size_t chunk = size_t(1) << 20;
            for (size_t off = 0; off < big_n; off += chunk) {
                hipMemcpyAsync(dbi + off * 32, bi + off * 32, chunk * 32, hipMemcpyHostToDevice, s0);
                hipMemcpyAsync(bo + off * 16, dbo + off * 16, chunk * 16, hipMemcpyDeviceToHost, s1);
            }
            double t2 = now();
            hipStreamSynchronize(s0); hipStreamSynchronize(s1);
            double t3 = now();
            hipHostUnregister(bi); hipHostUnregister(bo);
            double t4 = now();
            printf("(f) 16 Mi rays: register 512 + 256 MB %6.2f ms (%s, %s), enqueue %6.2f ms, copies both ways %6.2f ms, unregister %6.2f ms\n",
                   (t1 - t0) * 1e3, hipGetErrorString(e1), hipGetErrorString(e2), (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3);
        }
        std::free(bi); std::free(bo); hipFree(dbi); hipFree(dbo);
    }
    {   // do an upload and a download overlap (full duplex), issued on two streams from pinned memory?
        hipStream_t s1; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
        for (int rep = 0; rep < 3; ++rep) {
            double t0 = now();
            for (int k = 0; k < 8; ++k) {
                hipMemcpyAsync(d_in, pin, in_b, hipMemcpyHostToDevice, s0);
                hipMemcpyAsync(pin_out, d_out, out_b, hipMemcpyDeviceToHost, s1);
            }
            hipStreamSynchronize(s0); hipStreamSynchronize(s1);
            double t1 = now();
            for (int k = 0; k < 8; ++k) {
                hipMemcpyAsync(d_in, pin, in_b, hipMemcpyHostToDevice, s0);
                hipMemcpyAsync(pin_out, d_out, out_b, hipMemcpyDeviceToHost, s0);
            }
            hipStreamSynchronize(s0);
            double t2 = now();
            if (rep) printf("(e) 8 x (H2D 32 MB + D2H 16 MB): two streams %6.2f ms, one stream %6.2f ms (sum of the parts: %.2f)\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, 8 * 0.89);
        }
    }
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        std::memcpy(pin, in, in_b);
        double t1 = now();
        std::string s; s.assign(pin_out, out_b);
        double t2 = now();
        volatile float acc = 0;
        for (size_t i = 0; i < n; ++i) { float r[2]; std::memcpy(r, in + i * 32 + 24, 8); if (r[0] < 0.f || r[1] <= r[0]) acc = acc + 1; }
        double t3 = now();
        if (rep) printf("(d) 1-thread memcpy 32 MB %6.2f ms (%.1f GB/s); std::string::assign 16 MB from pinned %6.2f ms; range scan of 1 Mi rays %6.2f ms\n",
                        (t1 - t0) * 1e3, in_b / (t1 - t0) / 1e9, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
    }
    return 0;
}
