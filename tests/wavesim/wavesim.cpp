// wavesim.cpp -- fiber scheduler of the wave64 simulator (TEST INFRASTRUCTURE, see wavesim.h).
#include "wavesim.h"
#include <sys/mman.h>

namespace ws {
unsigned long long g_stat[32];

Lane* cur = nullptr;
Block* blk = nullptr;
void* sched_sp = nullptr;
dim3_ g_blockIdx, g_blockDim, g_gridDim;
static const std::function<void()>* g_body = nullptr;

// x86-64 SysV context switch: save callee-saved registers on the current stack, swap stack pointers.
asm(R"(
.text
.globl ws_switch
.type ws_switch,@function
ws_switch:
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
.size ws_switch,.-ws_switch
)");

static void lane_entry()
{
    (*g_body)();
    Lane& me = *cur;
    me.done = true;
    me.wave->live--;
    blk->live--;
    blk->progress++;
    ws_switch(&me.sp, sched_sp);
    abort();   // never resumed
}

static const size_t STACK = 256 * 1024;
static char* g_stacks = nullptr;
static size_t g_nstacks = 0;

void launch(dim3_ grid, dim3_ block, const std::function<void()>& body)
{
    const unsigned nthreads = block.x * block.y * block.z;
    if (nthreads > g_nstacks) {
        if (g_stacks) munmap(g_stacks, g_nstacks * STACK);
        g_stacks = (char*)mmap(nullptr, (size_t)nthreads * STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (g_stacks == (char*)MAP_FAILED) { perror("mmap"); abort(); }
        g_nstacks = nthreads;
    }
    g_body = &body;
    g_gridDim = grid; g_blockDim = block;
    for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
    for (unsigned bx = 0; bx < grid.x; bx++) {
        g_blockIdx = dim3_(bx, by, bz);
        Block B;
        B.lanes.resize(nthreads);
        B.waves.resize((nthreads + 63) / 64);
        B.live = (int)nthreads;
        blk = &B;
        for (unsigned t = 0; t < nthreads; t++) {
            Lane& L = B.lanes[t];
            L.tid = t;
            L.wave = &B.waves[t / 64];
            L.wave->live++;
            // initial stack: 6 callee-saved slots + return address (lane_entry); after `ret`, rsp % 16 == 8
            uintptr_t top = (uintptr_t)(g_stacks + (size_t)(t + 1) * STACK);
            top &= ~(uintptr_t)15;
            void** sp = (void**)top;
            *--sp = nullptr;                 // fake return address of lane_entry (alignment slot)
            *--sp = (void*)&lane_entry;      // popped by ret
            for (int k = 0; k < 6; k++) *--sp = nullptr;
            L.sp = (void*)sp;
        }
        unsigned long last_progress = ~0ul;
        int idle_passes = 0;
        while (B.live > 0) {
            for (unsigned t = 0; t < nthreads; t++) {
                Lane& L = B.lanes[t];
                if (L.done) continue;
                cur = &L;
                ws_switch(&sched_sp, L.sp);
            }
            if (B.progress == last_progress) {
                if (++idle_passes > 2) {
                    fprintf(stderr, "wavesim: deadlock in block (%u,%u,%u): divergent collective or barrier (block of %u threads, %d live, at the barrier: %d / %d)\n", bx, by, bz, nthreads, B.live, B.barrier_arrived[0], B.barrier_arrived[1]);
                    for (size_t w = 0; w < B.waves.size(); w++) fprintf(stderr, "  wave %zu: %d live, in a collective: %d / %d\n", w, B.waves[w].live, B.waves[w].arrived[0], B.waves[w].arrived[1]);
                    abort();
                }
            } else { idle_passes = 0; last_progress = B.progress; }
        }
        cur = nullptr; blk = nullptr;
    }
    g_body = nullptr;
}

}  // namespace ws
