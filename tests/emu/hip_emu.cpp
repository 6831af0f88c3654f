// TEST INFRASTRUCTURE ONLY — see hip_emu.h.
#include "hip_emu.h"

#include <ucontext.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

thread_local emu_uint3 threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;

namespace {
constexpr size_t kStack = 96 * 1024;
constexpr int kMaxThreads = 1024;

struct BlockCtx {
    ucontext_t main_ctx;
    ucontext_t fib[kMaxThreads];
    char* stacks = nullptr;
    bool done[kMaxThreads];
    int n = 0, cur = 0;
    // block barrier
    int arrived = 0;
    unsigned gen = 0;
    // wave collectives
    int w_arrived[kMaxThreads / 64];
    unsigned w_gen[kMaxThreads / 64];
    float w_val[kMaxThreads / 64][64];
    const std::function<void()>* body = nullptr;
};
thread_local BlockCtx* tls = nullptr;

void yield_to_main() {
    BlockCtx* c = tls;
    swapcontext(&c->fib[c->cur], &c->main_ctx);
}

void trampoline() {
    BlockCtx* c = tls;
    (*c->body)();
    c->done[c->cur] = true;
    swapcontext(&c->fib[c->cur], &c->main_ctx);
}

void set_tid(BlockCtx* c, int t) {
    c->cur = t;
    threadIdx.x = t % blockDim.x;
    threadIdx.y = (t / blockDim.x) % blockDim.y;
    threadIdx.z = t / (blockDim.x * blockDim.y);
}

void run_block(BlockCtx* c) {
    int n = c->n;
    c->arrived = 0;
    for (int w = 0; w < kMaxThreads / 64; ++w) c->w_arrived[w] = 0;
    for (int t = 0; t < n; ++t) {
        c->done[t] = false;
        getcontext(&c->fib[t]);
        c->fib[t].uc_stack.ss_sp = c->stacks + (size_t)t * kStack;
        c->fib[t].uc_stack.ss_size = kStack;
        c->fib[t].uc_link = nullptr;
        makecontext(&c->fib[t], trampoline, 0);
    }
    int alive = n;
    while (alive > 0) {
        alive = 0;
        for (int t = 0; t < n; ++t) {
            if (c->done[t]) continue;
            set_tid(c, t);
            swapcontext(&c->main_ctx, &c->fib[t]);
            if (!c->done[t]) ++alive;
        }
    }
}

int wave_lanes(BlockCtx* c, int w) {
    int rem = c->n - w * 64;
    return rem >= 64 ? 64 : rem;
}

void wave_barrier(BlockCtx* c, int w) {
    unsigned g = c->w_gen[w];
    if (++c->w_arrived[w] == wave_lanes(c, w)) {
        c->w_arrived[w] = 0;
        c->w_gen[w]++;
    } else {
        int me = c->cur;
        while (c->w_gen[w] == g) { yield_to_main(); set_tid(c, me); }
    }
}
}  // namespace

void emu_syncthreads() {
    BlockCtx* c = tls;
    unsigned g = c->gen;
    if (++c->arrived == c->n) {
        c->arrived = 0;
        c->gen++;
    } else {
        int me = c->cur;
        while (c->gen == g) { yield_to_main(); set_tid(c, me); }
    }
}

static float wave_collect(float v, int mode, int src) {
    BlockCtx* c = tls;
    int t = c->cur, w = t / 64, lane = t % 64;
    c->w_val[w][lane] = v;
    wave_barrier(c, w);
    int nl = wave_lanes(c, w);
    float r;
    if (mode == 0) { r = 0.f; for (int i = 0; i < nl; ++i) r += c->w_val[w][i]; }
    else if (mode == 1) { r = c->w_val[w][0]; for (int i = 1; i < nl; ++i) r = fmaxf(r, c->w_val[w][i]); }
    else r = c->w_val[w][src % nl];
    wave_barrier(c, w);
    return r;
}
float emu_wave_sum(float v) { return wave_collect(v, 0, 0); }
float emu_wave_max(float v) { return wave_collect(v, 1, 0); }
float emu_shfl(float v, int src_lane) { return wave_collect(v, 2, src_lane); }

float atomicAdd(float* p, float v) {
    auto* a = reinterpret_cast<std::atomic<uint32_t>*>(p);
    uint32_t old = a->load(std::memory_order_relaxed);
    for (;;) {
        float f; memcpy(&f, &old, 4);
        float nf = f + v; uint32_t nb; memcpy(&nb, &nf, 4);
        if (a->compare_exchange_weak(old, nb)) return f;
    }
}
int atomicAdd(int* p, int v) { return reinterpret_cast<std::atomic<int>*>(p)->fetch_add(v); }

hipError_t hipMalloc(void** p, size_t n) {
    void* q = nullptr;
    if (posix_memalign(&q, 256, n ? n : 256) != 0) return 1;
    *p = q;
    return 0;
}
hipError_t hipFree(void* p) { free(p); return 0; }

namespace {
struct Pool {
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::vector<std::thread> threads;
    const std::function<void()>* body = nullptr;
    dim3 grid, block;
    long nblocks = 0;
    std::atomic<long> next{0};
    int active = 0;
    unsigned long epoch = 0;
    bool stop = false;

    static void run_blocks(Pool* p) {
        static thread_local BlockCtx* ctx = nullptr;
        if (!ctx) { ctx = new BlockCtx(); ctx->stacks = (char*)malloc(kStack * kMaxThreads); }
        tls = ctx;
        ctx->n = (int)(p->block.x * p->block.y * p->block.z);
        ctx->body = p->body;
        blockDim = p->block;
        gridDim = p->grid;
        for (;;) {
            long b = p->next.fetch_add(1);
            if (b >= p->nblocks) break;
            blockIdx.x = (unsigned)(b % p->grid.x);
            blockIdx.y = (unsigned)((b / p->grid.x) % p->grid.y);
            blockIdx.z = (unsigned)(b / ((long)p->grid.x * p->grid.y));
            run_block(ctx);
        }
    }
    void loop() {
        unsigned long seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            cv_job.wait(lk, [&] { return stop || epoch != seen; });
            if (stop) return;
            seen = epoch;
            lk.unlock();
            run_blocks(this);
            lk.lock();
            if (--active == 0) cv_done.notify_all();
        }
    }
    explicit Pool(int n) { for (int i = 0; i < n; ++i) threads.emplace_back([this] { loop(); }); }
    std::mutex launch_mu;  // one kernel at a time (host threads driving different handles queue up, like on one GPU)
    void launch(dim3 g, dim3 b, const std::function<void()>& f) {
        std::lock_guard<std::mutex> one(launch_mu);
        std::unique_lock<std::mutex> lk(mu);
        grid = g; block = b; body = &f;
        nblocks = (long)g.x * g.y * g.z;
        next = 0;
        active = (int)threads.size();
        ++epoch;
        cv_job.notify_all();
        cv_done.wait(lk, [&] { return active == 0; });
    }
};
}  // namespace

void emu_launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads > kMaxThreads) { fprintf(stderr, "emu: block too large\n"); abort(); }
    static Pool* pool = [] {
        const char* e = getenv("MTTS_EMU_THREADS");
        int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
        return new Pool(n < 1 ? 1 : n);
    }();
    pool->launch(grid, block, body);
}
