// TEST INFRASTRUCTURE: fiber scheduler of the SIMT emulator + the product kernels compiled for the
// host.  Build: see tests/emu/build_emu.py.
#include <hip/hip_runtime.h>
#include <thread>
#include <vector>
#include <atomic>
#include <mutex>

namespace emu {
thread_local Block *tb = nullptr;

extern "C" void emu_switch(void **from_sp, void *to_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
)");

static const size_t kStack = 256 * 1024;

static void fiber_entry() {
    Block *b = tb;
    (*b->body)();
    b->fibers[b->cur].done = true;
    emu_switch(&b->fibers[b->cur].sp, b->sched_sp);
    abort();
}

void yield() {
    Block *b = tb;
    emu_switch(&b->fibers[b->cur].sp, b->sched_sp);
}

void block_barrier() {
    Block *b = tb;
    const unsigned long g = b->block_gen;
    if (++b->block_arrived == b->nthreads) { b->block_arrived = 0; b->block_gen++; return; }
    while (b->block_gen == g) yield();
}

void wave_barrier() {
    Block *b = tb;
    const int w = wave();
    const int n = (b->nthreads - w * 64) < 64 ? (b->nthreads - w * 64) : 64;
    const unsigned long g = b->wave_gen[w];
    if (++b->wave_arrived[w] == n) { b->wave_arrived[w] = 0; b->wave_gen[w]++; return; }
    while (b->wave_gen[w] == g) yield();
}

static void run_block(Block *b) {
    for (int t = 0; t < b->nthreads; ++t) {
        Fiber &f = b->fibers[t];
        f.done = false;
        f.tidx = {(unsigned)t, 0u, 0u};
        uintptr_t top = ((uintptr_t)(f.stack + kStack)) & ~(uintptr_t)15;
        void **sp = (void **)top;
        *--sp = nullptr;                       // fake return address of fiber_entry
        *--sp = (void *)&fiber_entry;          // 'ret' target
        for (int i = 0; i < 6; ++i) *--sp = nullptr;
        f.sp = sp;
    }
    b->block_gen = 0; b->block_arrived = 0;
    memset(b->wave_gen, 0, sizeof(b->wave_gen));
    memset(b->wave_arrived, 0, sizeof(b->wave_arrived));
    memset(b->wave_op, 0, sizeof(b->wave_op));
    int alive = b->nthreads;
    while (alive > 0) {
        for (int t = 0; t < b->nthreads; ++t) {
            if (b->fibers[t].done) continue;
            b->cur = t;
            emu_switch(&b->sched_sp, b->fibers[t].sp);
            if (b->fibers[t].done) --alive;
        }
    }
}

// Fiber stacks are recycled across launches: a launch used to malloc / free 256 KB per emulated thread and worker (an mmap +
// munmap + page faults each: ~20 ms per launch with 512-thread workgroups on 8 workers, more than the kernels themselves for the
// ~1 700 small launches of a ResNet-101 training step).
static std::mutex g_stack_mu;
static std::vector<char *> g_stack_pool;
static char *stack_get() {
    {
        std::lock_guard<std::mutex> lock(g_stack_mu);
        if (!g_stack_pool.empty()) { char *s = g_stack_pool.back(); g_stack_pool.pop_back(); return s; }
    }
    return (char *)malloc(kStack);
}
static void stack_put(char *s) {
    std::lock_guard<std::mutex> lock(g_stack_mu);
    if (g_stack_pool.size() < 8192) g_stack_pool.push_back(s);
    else free(s);
}

void launch(const std::function<void()> &body, dim3 grid, dim3 block, size_t shmem) {
    const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    const int nthreads = (int)block.x;
    unsigned hw = std::thread::hardware_concurrency();
    size_t nworkers = hw ? hw : 4;
    if (nworkers > nblocks) nworkers = nblocks;
    std::atomic<size_t> next(0);
    auto worker = [&]() {
        Block *b = new Block();
        b->fibers = new Fiber[nthreads];
        for (int t = 0; t < nthreads; ++t) b->fibers[t].stack = stack_get();
        b->nthreads = nthreads;
        b->bdim = block; b->gdim = grid;
        b->dyn_lds = (char *)aligned_alloc(64, ((shmem + 63) / 64 + 1) * 64);
        b->body = &body;
        tb = b;
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= nblocks) break;
            b->bidx = dim3((unsigned)(i % grid.x), (unsigned)((i / grid.x) % grid.y), (unsigned)(i / ((size_t)grid.x * grid.y)));
            memset(b->dyn_lds, 0xFF, shmem);      // poison: uninitialised LDS reads show up as NaNs
            run_block(b);
        }
        for (int t = 0; t < nthreads; ++t) stack_put(b->fibers[t].stack);
        delete[] b->fibers;
        free(b->dyn_lds);
        delete b;
        tb = nullptr;
    };
    std::vector<std::thread> th;
    for (size_t w = 0; w < nworkers; ++w) th.emplace_back(worker);
    for (auto &t : th) t.join();
}
}  // namespace emu

