/* oracle/shim/simt_emu.cpp — scheduler of the SIMT emulator (see simt_emu.h).  TEST INFRASTRUCTURE ONLY. */
#include "simt_emu.h"

namespace gs_emu {
Fiber* cur = nullptr;
gs_dim3 bIdx, bDim, gDim;
static ucontext_t sched_ctx;
static const std::function<void()>* g_body = nullptr;
static const size_t STACK_BYTES = 256 * 1024;

static void park(State s) {
    cur->state = s;
    swapcontext(&cur->ctx, &sched_ctx);
}
uint64_t collective(Op op, uint32_t mask, uint64_t val, int arg) {
    cur->op = op;
    cur->mask = mask;
    cur->val = val;
    cur->arg = arg;
    park(WAIT_WARP);
    return cur->result;
}
void barrier() { park(WAIT_BLOCK); }

static void trampoline() {
    (*g_body)();
    cur->state = DONE;
    swapcontext(&cur->ctx, &sched_ctx);
}
[[noreturn]] static void die(const char* what) {
    fprintf(stderr, "simt_emu: %s (block %u)\n", what, bIdx.x);
    abort();
}

/* complete one collective for the lanes in `part` (bit l = lane l) of the warp starting at fibers[base] */
static void resolve(std::vector<Fiber>& f, unsigned base, unsigned nl, uint32_t part) {
    const Fiber& first = f[base + (unsigned)__builtin_ctz(part)];
    const Op op = first.op;
    uint32_t ballot = 0;
    for (unsigned l = 0; l < nl; ++l)
        if (((part >> l) & 1u) && f[base + l].val && op == OP_BALLOT) ballot |= 1u << l;
    for (unsigned l = 0; l < nl; ++l) {
        if (!((part >> l) & 1u)) continue;
        Fiber& me = f[base + l];
        int src = (int)l;
        switch (op) {
        case OP_BALLOT: me.result = ballot; break;
        case OP_ACTIVEMASK: me.result = part; break;
        case OP_SHFL: src = me.arg & 31; break;
        case OP_SHFL_UP: src = (int)l - me.arg; break;
        case OP_SHFL_DOWN: src = (int)l + me.arg; break;
        case OP_SHFL_XOR: src = (int)l ^ me.arg; break;
        default: die("unknown collective");
        }
        if (op != OP_BALLOT && op != OP_ACTIVEMASK)  // out of range / inactive source lane: the caller's own value
            me.result = (src >= 0 && src < (int)nl && ((part >> src) & 1u)) ? f[base + (unsigned)src].val : me.val;
    }
    for (unsigned l = 0; l < nl; ++l)
        if ((part >> l) & 1u) f[base + l].state = RUNNABLE;
}

void launch(unsigned grid, unsigned block, const std::function<void()>& body) {
    g_body = &body;
    gDim = {grid, 1, 1};
    bDim = {block, 1, 1};
    std::vector<Fiber> f(block);
    for (unsigned t = 0; t < block; ++t) f[t].stack = malloc(STACK_BYTES);
    for (unsigned b = 0; b < grid; ++b) {
        bIdx = {b, 0, 0};
        for (unsigned t = 0; t < block; ++t) {
            Fiber& x = f[t];
            x.tid = {t, 0, 0};
            x.lane = t & 31u;
            x.warp = t >> 5;
            x.state = RUNNABLE;
            x.op = OP_NONE;
            getcontext(&x.ctx);
            x.ctx.uc_stack.ss_sp = x.stack;
            x.ctx.uc_stack.ss_size = STACK_BYTES;
            x.ctx.uc_link = nullptr;
            makecontext(&x.ctx, trampoline, 0);
        }
        for (;;) {
            for (unsigned t = 0; t < block; ++t)
                if (f[t].state == RUNNABLE) {
                    cur = &f[t];
                    swapcontext(&sched_ctx, &f[t].ctx);
                }
            bool progressed = false;
            for (unsigned base = 0; base < block; base += 32) {
                const unsigned nl = block - base < 32 ? block - base : 32;
                uint32_t waiting = 0;
                for (unsigned l = 0; l < nl; ++l)
                    if (f[base + l].state == WAIT_WARP) waiting |= 1u << l;
                while (waiting) {
                    const unsigned l0 = (unsigned)__builtin_ctz(waiting);
                    const Fiber& a = f[base + l0];
                    uint32_t part = 0;
                    if (a.op == OP_ACTIVEMASK) {
                        for (unsigned l = 0; l < nl; ++l)
                            if (((waiting >> l) & 1u) && f[base + l].op == OP_ACTIVEMASK) part |= 1u << l;
                    } else {
                        part = a.mask & (nl == 32 ? 0xffffffffu : ((1u << nl) - 1u));
                        bool ready = (part >> l0) & 1u;
                        for (unsigned l = 0; l < nl && ready; ++l)
                            if ((part >> l) & 1u)
                                ready = f[base + l].state == WAIT_WARP && f[base + l].op == a.op && f[base + l].mask == a.mask;
                        if (!ready) {  // the other lanes of the mask have not arrived (they cannot run any more: deadlock)
                            waiting &= ~(1u << l0);
                            continue;
                        }
                    }
                    resolve(f, base, nl, part);
                    waiting &= ~part;
                    progressed = true;
                }
            }
            if (progressed) continue;
            unsigned at_barrier = 0, done = 0, stuck = 0;
            for (unsigned t = 0; t < block; ++t) {
                at_barrier += f[t].state == WAIT_BLOCK;
                done += f[t].state == DONE;
                stuck += f[t].state == WAIT_WARP;
            }
            if (stuck) die("warp collective whose participants never arrive");
            if (done == block) break;
            if (at_barrier + done != block) die("scheduler inconsistency");
            for (unsigned t = 0; t < block; ++t)
                if (f[t].state == WAIT_BLOCK) f[t].state = RUNNABLE;  // CUDA: exited threads do not take part
        }
    }
    for (unsigned t = 0; t < block; ++t) free(f[t].stack);
    cur = nullptr;
}
}  // namespace gs_emu
