// vo_knn2_hamming_static: the Hamming kNN-2 search of vo_knn2_hamming for
// callers that cannot allocate device memory (code under HIP graph capture,
// serving loops, harnesses that forbid hipMalloc).  Same kernels; everything
// they need beyond the caller's inputs and outputs -- the two descriptor sets
// in the slot layout the kernels expect, the per-split partial top-2 lists, the
// merged records and the few control words -- lives in one file-scope device
// array.  No context, no hipMalloc / hipFree / stream or event creation.
//
// Every GPU holds its own copy of the array at its own address.  Host code must
// never hand the array's *name* to a kernel or a copy (on the host the name is a
// stand-in, not a device address): the device address is looked up once per
// device with hipGetSymbolAddress and only that pointer is used afterwards.
#include <stdio.h>

#include <mutex>

#include "vo_internal.h"

using namespace vo;

namespace {

constexpr int kStaticMax = VO_KNN2_STATIC_MAX;  // descriptors per set
constexpr int kMaxDevices = 64;

// layout of the scratch array (byte offsets, all multiples of 256)
constexpr size_t kOffCtl = 0;                                         // nkp[2], q_slot, t_slot, n_good
constexpr size_t kOffDesc = 256;                                      // 2 slots x kStaticMax x 32 B
constexpr size_t kOffPartial = kOffDesc + (size_t)2 * kStaticMax * 32;            // kKnnSplits x kStaticMax x uint2
constexpr size_t kOffKnn = kOffPartial + (size_t)kKnnSplits * kStaticMax * sizeof(uint2);  // kStaticMax x vo_knn2
constexpr size_t kScratchBytes = kOffKnn + (size_t)kStaticMax * sizeof(vo_knn2);
static_assert(kOffPartial % 256 == 0 && kOffKnn % 256 == 0, "sub-buffers stay 256-byte aligned");
static_assert(kScratchBytes <= (1u << 20), "the static scratch stays modest (<= 1 MiB per GPU)");

}  // namespace

// The scratch itself: held by every GPU the library runs on until the process ends.
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

namespace {

std::mutex g_mu;
unsigned char *g_ptr[kMaxDevices] = {};  // device address of `scratch` on each GPU (null: not looked up yet)
thread_local char g_err[256] = "";

// control words: the two sets sit in slots 0 (query) and 1 (train)
__global__ void k_static_control(int *ctl, int nq, int nt)
{
    if (threadIdx.x == 0) {
        ctl[0] = nq;  // nkp[slot 0]
        ctl[1] = nt;  // nkp[slot 1]
        ctl[2] = 0;   // q_slot of pair 0
        ctl[3] = 1;   // t_slot of pair 0
        ctl[4] = 0;   // n_good (written by the merge kernel as well)
    }
}

int fail(int code, const char *what, hipError_t e)
{
    snprintf(g_err, sizeof(g_err), "vo_knn2_hamming_static: %s%s%s", what, e != hipSuccess ? ": " : "",
             e != hipSuccess ? hipGetErrorString(e) : "");
    return code;
}

// device address of the scratch array on `device` (which must be current)
int scratch_of(int device, unsigned char **out)
{
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_ptr[device]) {
        void *p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) return fail(VO_ERR_HIP, "hipGetSymbolAddress(scratch)", e);
        g_ptr[device] = static_cast<unsigned char *>(p);
    }
    *out = g_ptr[device];
    return VO_OK;
}

}  // namespace

extern "C" {

const char *vo_static_last_error(void) { return g_err; }

int vo_knn2_hamming_static(int device, void *hip_stream, const uint8_t *query, int nq, const uint8_t *train, int nt,
                           vo_knn2 *out, int wait)
{
    g_err[0] = 0;
    if (device < 0 || device >= kMaxDevices || nq < 0 || nt < 0 || (nq > 0 && (!query || !out)) || (nt > 0 && !train))
        return fail(VO_ERR_ARG, "bad arguments", hipSuccess);
    if (nq > kStaticMax || nt > kStaticMax) {
        snprintf(g_err, sizeof(g_err), "vo_knn2_hamming_static: %d queries / %d trains exceed the static limit of %d "
                 "descriptors per set (use vo_knn2_hamming with a context for larger sets)", nq, nt, kStaticMax);
        return VO_ERR_ARG;
    }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return fail(VO_ERR_HIP, "hipSetDevice", e);
    unsigned char *base = nullptr;
    int rc = scratch_of(device, &base);  // (also what a warm-up call with n_query = 0 is for)
    if (rc) return rc;
    if (nq == 0) return VO_OK;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);

    int *ctl = reinterpret_cast<int *>(base + kOffCtl);
    uint8_t *desc = base + kOffDesc;
    uint2 *partial = reinterpret_cast<uint2 *>(base + kOffPartial);
    vo_knn2 *knn = reinterpret_cast<vo_knn2 *>(base + kOffKnn);
    const size_t slot_bytes = (size_t)kStaticMax * 32;

    // Nothing in the scratch has to start at zero: the control words are rewritten here, every split writes the
    // partial record of every query below nq, and the merge writes records 0 .. nq-1 -- the only ones copied out.
    hipLaunchKernelGGL(k_static_control, dim3(1), dim3(64), 0, s, ctl, nq, nt);
    if ((e = hipMemcpyAsync(desc, query, (size_t)nq * 32, hipMemcpyDefault, s)) != hipSuccess)
        return fail(VO_ERR_HIP, "copy of the query descriptors", e);
    if (nt > 0 && (e = hipMemcpyAsync(desc + slot_bytes, train, (size_t)nt * 32, hipMemcpyDefault, s)) != hipSuccess)
        return fail(VO_ERR_HIP, "copy of the train descriptors", e);
    launch_knn2(desc, slot_bytes, ctl, ctl + 2, ctl + 3, kStaticMax, partial, nq, 1, kKnnSplits, false, s);
    // ratio 0 keeps nothing, so the merge kernel touches neither key points nor a match list
    launch_match_finalize(partial, nullptr, kStaticMax, ctl, ctl + 2, ctl + 3, 0.0f, nullptr, knn, nullptr, ctl + 4, 1,
                          kKnnSplits, s);
    if ((e = hipGetLastError()) != hipSuccess) return fail(VO_ERR_HIP, "kernel launch", e);
    if ((e = hipMemcpyAsync(out, knn, sizeof(vo_knn2) * (size_t)nq, hipMemcpyDefault, s)) != hipSuccess)
        return fail(VO_ERR_HIP, "copy of the result", e);
    if (wait && (e = hipStreamSynchronize(s)) != hipSuccess) return fail(VO_ERR_HIP, "hipStreamSynchronize", e);
    return VO_OK;
}

}  // extern "C"
