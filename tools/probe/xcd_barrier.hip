// Grid barriers on MI355X, measured (development probe; results in profiles/r04_grid_barrier.txt).
//   flat   one device-scope counter + generation word: every block arrives with an atomic, polls the generation with relaxed
//          sc1 loads + s_sleep; release fence before the arrival, acquire fence after the wait
//   xcd    hierarchical: per-XCC arrival counter; the last arriver of an XCC arrives at the top counter; the last XCC publishes the
//          new generation to the eight per-XCC generation words that the blocks of that XCC poll
// One block per CU (256 or 512 blocks of 256 threads), `iters` barriers in a loop; optional payload: every block writes a 128-byte
// record before the barrier and reads its right neighbour's record after it (checked).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Bar {
    unsigned int flat_cnt; unsigned int pad0[31];
    unsigned int flat_gen; unsigned int pad1[31];
    unsigned int top_cnt; unsigned int pad2[31];
    unsigned int xcc_cnt[8][32];       // one 128-byte line each
    unsigned int xcc_gen[8][32];
    unsigned int xcc_members[8][32];
};

__device__ inline unsigned int xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u; }   // HW_REG_XCC_ID[3:0]

__device__ inline unsigned int ld_relaxed(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int FENCES = 1>
__device__ void barrier_flat(Bar* b, unsigned int gen, unsigned int nblocks) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (FENCES) __atomic_thread_fence(__ATOMIC_RELEASE);   // agent scope by default for device code
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int t = __hip_atomic_fetch_add(&b->flat_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == nblocks - 1) {
            __hip_atomic_store(&b->flat_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (FENCES) __hip_atomic_store(&b->flat_gen, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_store(&b->flat_gen, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (ld_relaxed(&b->flat_gen) != gen) __builtin_amdgcn_s_sleep(1);
        }
        if (FENCES) __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
}

template <int FENCES = 1>
__device__ void barrier_xcd(Bar* b, unsigned int gen, unsigned int x, unsigned int members, unsigned int n_xcc) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (!FENCES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int t = __hip_atomic_fetch_add(&b->xcc_cnt[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == members - 1) {                                  // XCC leader: last arriver of this XCC
            __hip_atomic_store(&b->xcc_cnt[x][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (FENCES) __atomic_thread_fence(__ATOMIC_RELEASE);
            const unsigned int u = __hip_atomic_fetch_add(&b->top_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (u == n_xcc - 1) {
                __hip_atomic_store(&b->top_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (FENCES) __atomic_thread_fence(__ATOMIC_ACQUIRE);
                for (unsigned int k = 0; k < n_xcc; ++k) __hip_atomic_store(&b->xcc_gen[k][0], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        while (ld_relaxed(&b->xcc_gen[x][0]) != gen) __builtin_amdgcn_s_sleep(1);
        if (FENCES) __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
}

// XCC-local: only the blocks that share one L2 meet (per-XCC counter + generation word, nothing chip-wide).  FENCES = 1: agent-scope
// release / acquire fences around it (L2 write-back + invalidate, what a chip-wide hand-off needs); FENCES = 0: none -- the payload
// travels in sc1 stores / loads that are performed at the shared L2 anyway, the wave only waits for its own stores to complete.
template <int FENCES>
__device__ void barrier_local(Bar* b, unsigned int gen, unsigned int x, unsigned int members) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (FENCES) __atomic_thread_fence(__ATOMIC_RELEASE); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int t = __hip_atomic_fetch_add(&b->xcc_cnt[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == members - 1) {
            __hip_atomic_store(&b->xcc_cnt[x][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&b->xcc_gen[x][0], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (ld_relaxed(&b->xcc_gen[x][0]) != gen) __builtin_amdgcn_s_sleep(1);
        }
        if (FENCES) __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void count_members(Bar* b) {
    if (threadIdx.x == 0) atomicAdd(&b->xcc_members[xcc_id()][0], 1u);
}

template <int KIND, int PAYLOAD>
__global__ __launch_bounds__(256) void bench_kernel(Bar* b, unsigned int gen0, int iters, unsigned int* records, unsigned int* errors) {
    const unsigned int nb = gridDim.x, me = blockIdx.x, x = xcc_id();
    const unsigned int members = b->xcc_members[x][0];
    unsigned int n_xcc = 0;
    for (int k = 0; k < 8; ++k) n_xcc += b->xcc_members[k][0] ? 1u : 0u;
    unsigned int bad = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned int gen = gen0 + it + 1;
        if (PAYLOAD && threadIdx.x < 32) __hip_atomic_store(records + (size_t)me * 32 + threadIdx.x, gen * 1000u + me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (KIND == 0) barrier_flat(b, gen, nb); else if (KIND == 1) barrier_xcd(b, gen, x, members, n_xcc); else if (KIND == 4) barrier_flat<0>(b, gen, nb); else if (KIND == 5) barrier_xcd<0>(b, gen, x, members, n_xcc); else barrier_local<KIND == 2>(b, gen, x, members);
        if (PAYLOAD && threadIdx.x < 32) {
            const unsigned int nbr = (KIND == 2 || KIND == 3) ? (me + 8) % nb : (me + 1) % nb;   // XCC-local: a block of the same XCC (blocks are dealt round-robin)
            const unsigned int v = __hip_atomic_load(records + (size_t)nbr * 32 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != gen * 1000u + nbr) ++bad;
        }
        if (PAYLOAD) {                                            // nobody overwrites a record before its reader has it
            if (KIND == 0) barrier_flat(b, gen + 1000000u, nb); else if (KIND == 1) barrier_xcd(b, gen + 1000000u, x, members, n_xcc); else if (KIND == 4) barrier_flat<0>(b, gen + 1000000u, nb); else if (KIND == 5) barrier_xcd<0>(b, gen + 1000000u, x, members, n_xcc); else barrier_local<KIND == 2>(b, gen + 1000000u, x, members);
        }
    }
    if (bad) atomicAdd(errors, bad);
}

template <int KIND, int PAYLOAD>
static double run(Bar* d_bar, int blocks, int iters, unsigned int* rec, unsigned int* err, unsigned int& gen) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((bench_kernel<KIND, PAYLOAD>), dim3(blocks), dim3(256), 0, 0, d_bar, gen, iters, rec, err);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        gen += 2000000u;
        if (rep > 0 && ms < best) best = ms;
    }
    return best * 1e3 / iters / (PAYLOAD ? 2 : 1);
}

int main() {
    Bar* d_bar; unsigned int *rec, *err;
    CHECK(hipMalloc(&d_bar, sizeof(Bar))); CHECK(hipMalloc(&rec, 2048 * 128)); CHECK(hipMalloc(&err, 4));
    for (int blocks : {256, 512}) {
        CHECK(hipMemset(d_bar, 0, sizeof(Bar))); CHECK(hipMemset(err, 0, 4));
        hipLaunchKernelGGL(count_members, dim3(blocks), dim3(256), 0, 0, d_bar);
        CHECK(hipDeviceSynchronize());
        Bar h; CHECK(hipMemcpy(&h, d_bar, sizeof(Bar), hipMemcpyDeviceToHost));
        printf("%d blocks: members per XCC =", blocks);
        for (int k = 0; k < 8; ++k) printf(" %u", h.xcc_members[k][0]);
        printf("\n");
        unsigned int gen = 0;
        const int iters = 200;
        const double f0 = run<0, 0>(d_bar, blocks, iters, rec, err, gen), x0 = run<1, 0>(d_bar, blocks, iters, rec, err, gen);
        const double f1 = run<0, 1>(d_bar, blocks, iters, rec, err, gen), x1 = run<1, 1>(d_bar, blocks, iters, rec, err, gen);
        unsigned int e = 0; CHECK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
        printf("  us per barrier: flat %.2f  xcd %.2f | with a 128-byte record per block (write, barrier, read neighbour): flat %.2f  xcd %.2f | record errors %u\n", f0, x0, f1, x1, e);
        CHECK(hipMemset(err, 0, 4));
        const double l2 = run<2, 0>(d_bar, blocks, iters, rec, err, gen), l3 = run<3, 0>(d_bar, blocks, iters, rec, err, gen);
        const double l2p = run<2, 1>(d_bar, blocks, iters, rec, err, gen), l3p = run<3, 1>(d_bar, blocks, iters, rec, err, gen);
        CHECK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
        CHECK(hipMemset(err, 0, 4));
        const double f4 = run<4, 0>(d_bar, blocks, iters, rec, err, gen), x5 = run<5, 0>(d_bar, blocks, iters, rec, err, gen);
        const double f4p = run<4, 1>(d_bar, blocks, iters, rec, err, gen), x5p = run<5, 1>(d_bar, blocks, iters, rec, err, gen);
        unsigned int e2 = 0; CHECK(hipMemcpy(&e2, err, 4, hipMemcpyDeviceToHost));
        printf("  chip-wide WITHOUT fences (payload in sc1 stores / loads): flat %.2f  xcd %.2f | with record: flat %.2f  xcd %.2f | record errors %u\n", f4, x5, f4p, x5p, e2);
        printf("  XCC-local (eight independent groups): agent fences %.2f  no fences %.2f | with record: agent fences %.2f  no fences %.2f | record errors %u\n", l2, l3, l2p, l3p, e);
    }
    return 0;
}
