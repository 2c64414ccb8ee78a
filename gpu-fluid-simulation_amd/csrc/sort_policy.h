// sort_policy.h — host side of the sort's late-stage plan (kernels_sort.hip, k_late_cert), shared by the 2D and the
// 3D engine.
//
// The device-side certificate of every sort reports the stage it ran at, its verdict and how much room the step's
// moves left ("fit class") into pinned host words.  The host reads them WITHOUT synchronising and picks the launch
// sequence of the next steps from them; at most FLIGHT steps are ever queued ahead of the device so that the reports
// are a few steps old.  Only the launch sequence depends on any of this — the kernels produce the reference network's
// arrangement whatever the plan, so a stale or wrong guess costs time, never a result.
//   - a failed certificate: two stages up, and the per-stage launches stand by again until reports pass;
//   - passed, but only with the full window (fit class 0: less than 2x room): one stage up;
//   - passed inside a quarter of the window four reports in a row: one stage down;
//   - the single stand-by launch replaces the per-stage ones once two reports in a row passed (a failure or an
//     upload resets that).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "fs_kernels.h"

namespace fsd {

struct SortPolicy {
    static const uint32_t FLIGHT = 4;
    uint32_t* fb = nullptr;         // pinned, mapped: [0] seq, [1] stage, [2] verdict, [3] fit class, [4] stand-by time-outs
    uint32_t seq = 0, seen = 0;
    uint32_t skip_seq = 1;          // the report of the first step after create / an upload says nothing about the flow
    int stage = 0;                  // 0: not chosen yet
    int roomy = 0;                  // consecutive reports with the moves inside a quarter of the window
    int trusted = 0;                // consecutive passing reports
    bool enabled = false;
    bool force_single = false;      // FS_SORT_TRUST=1 (tests): the single stand-by launch from the first step on
    bool inject_timeout = false;    // FS_SORT_INJECT_TIMEOUT=1 (tests): drives the host's handling of a barrier time-out
    int fixed_stage = -1;           // FS_SORT_FUSE_STAGE at create: a fixed stage (0: per-stage launches only), no policy
    int start_back = 8;             // first guess: S - start_back
    hipEvent_t flight[FLIGHT] = {};
    uint64_t enqueued = 0;

    hipError_t init(int start_back_) {
        start_back = start_back_;
        const char* e = getenv("FS_SORT_POLICY");
        const char* fx = getenv("FS_SORT_FUSE_STAGE");
        const char* tr = getenv("FS_SORT_TRUST");
        if (fx) fixed_stage = atoi(fx);
        force_single = tr && atoi(tr) != 0;
        { const char* inj = getenv("FS_SORT_INJECT_TIMEOUT"); inject_timeout = inj && atoi(inj) != 0; }
        { const char* q = getenv("FS_FORCE_QUAD_MAX"); if (q) quad_max = (uint32_t)atoi(q); }
        { const char* q = getenv("FS_FORCE_QUAD_MIN"); if (q) quad_min = (uint32_t)atoi(q); }
        { const char* q = getenv("FS_FORCE_QUAD_ALWAYS"); quad_always = q && atoi(q) != 0; }
        enabled = !(e && atoi(e) == 0) && !fx;
        // the words are allocated even when the plan policy is off: fb[6] is the force pass's work report (general_grid())
        hipError_t r = hipHostMalloc((void**)&fb, 8 * sizeof(uint32_t), hipHostMallocMapped);
        if (r == hipSuccess) memset(fb, 0, 8 * sizeof(uint32_t));
        return r;
    }
    void release() {
        for (auto& e : flight) { if (e) (void)hipEventDestroy(e); e = nullptr; }
        if (fb) (void)hipHostFree(fb);
        fb = nullptr;
    }
    // the state was replaced from outside: an arbitrary order
    void touched() { trusted = 0; skip_seq = seq + 1; }

    // Wait until the step enqueued FLIGHT steps ago has finished (call before enqueueing a step) ...
    hipError_t throttle() {
        hipEvent_t& slot = flight[enqueued % FLIGHT];
        if (slot) return hipEventSynchronize(slot);
        // only ever waited for by the host, which reads nothing the step wrote through it: no system-scope fence
        return hipEventCreateWithFlags(&slot, hipEventDisableTiming | hipEventDisableSystemFence);
    }
    // ... and mark the end of the step just enqueued: either the step's last kernel carried flight_event() as its completion
    // signal (hipExtLaunchKernelGGL: no marker packet — a hipEventRecord costs the stream ~5.7 us per step, 3 % of a 1 M-particle
    // step) and step_bound() is called, or step_enqueued() records it.
    hipEvent_t flight_event() const { return flight[enqueued % FLIGHT]; }
    void step_bound() { enqueued += 1; }
    hipError_t step_enqueued(hipStream_t st) {
        hipError_t r = hipEventRecord(flight[enqueued % FLIGHT], st);
        enqueued += 1;
        return r;
    }

    // One report of the certificate: sequence number, the stage it ran at, its verdict, the fit class.
    void observe(uint32_t s, int st, bool passed, int cls, uint32_t S) {
        seen = s;
        if (stage == 0) stage = st;
        if (s == skip_seq) {
            // nothing to learn
        } else if (!passed) {
            if (st + 2 > stage) stage = st + 2;
            trusted = 0; roomy = 0;
        } else if (cls == 0) {
            if (st + 1 > stage) stage = st + 1;     // a wider window is safer: the trust stays
            trusted += 1; roomy = 0;
        } else {
            trusted += 1;
            roomy = (cls >= 2 && st == stage) ? roomy + 1 : 0;
            // inside a quarter of the window: one stage down still leaves 2x room, the trust stays
            if (roomy >= 4 && stage > 13) { stage -= 1; roomy = 0; }
        }
        if (stage > (int)S - 1) stage = (int)S - 1;
    }
    int first_stage(uint32_t S) const { return (int)S - start_back < 13 ? 13 : (int)S - start_back; }
    bool single_standby() const { return force_single || (stage && trusted >= 2 && seq - seen <= 2 * FLIGHT); }

    // Workgroups for the force pass's general kernel (k_force_general) from its own report of a few steps ago: an idle
    // launch costs what its workgroups cost to come and go (16M: ~13 us with 1024, ~25 us with 4096), a busy one wants
    // them all (dense floor: force 1.175 ms with 1024, 1.117 with 4096).  Performance only.
    uint32_t general_grid() const {
        if (!fb) return 0;
        const uint32_t entries = ((const volatile uint32_t*)fb)[6];
        return entries == 0 ? 240u : entries < 128u ? 1040u : 4080u;      // multiples of 40: k_force_general's entry mapping
    }
    uint32_t* general_hint() const { return fb ? fb + 6 : nullptr; }
    // ... and whether that list is short enough for k_force_quad (the latency case: a small scene, a slab rank, the first dense
    // clusters); 0: no.  Lists of fewer than quad_min blocks are, as a rule, blocks that merely missed the LDS tile (a slab rank's
    // sparse ghost columns, a particle of spray: light waves): for those the quad kernel's chain of global round trips plus the
    // extra launch is SLOWER than the general kernel (one unfit block 13 + 4 us against 16; one rank of 8 0.218 -> 0.229 ms with
    // a bound of 8).  Telling heavy waves from light ones on the device was tried and cost more than it gave: one more
    // same-address atomic per registered wave cost k_density 5 % in the dense regime, a per-wave counter in k_force_general
    // cost that kernel its last free register (scratch, +4 %).  Performance only: every choice gives the same bits.
    uint32_t quad_max = 512, quad_min = 64;  // FS_FORCE_QUAD_MAX / _MIN (blocks; max 0 disables), read when the handle is created
    bool quad_always = false;                // FS_FORCE_QUAD_ALWAYS=1 (tests): k_force_quad in every step, whatever the list held
    uint32_t quad_entries() const {
        if (!fb) return 0;
        const uint32_t entries = ((const volatile uint32_t*)fb)[6];
        if (quad_always) return entries > 64u ? entries : 64u;
        return entries >= quad_min && entries <= quad_max ? entries : 0u;
    }

    // After a synchronisation of the simulation's stream: did the stand-by kernel (k_late_fallback) report a grid-barrier
    // time-out in any of the steps enqueued so far?  From that step on the particle order is undefined (include/fluidsim.h), so
    // every call that hands state to the caller checks this — not only the plan() of a later step.  `dirty`: the sort's tile
    // flags of `n` elements (the plan words behind them hold the count).  Latches.
    bool dead = false;
    hipError_t check_timeout(const uint32_t* dirty, uint32_t n) {
        if (dead || !enabled || n < (1u << 15)) return hipSuccess;
        uint32_t t = 0;
        const hipError_t r = hipMemcpy(&t, dirty + sort_plan_word(n) + 4, sizeof t, hipMemcpyDeviceToHost);
        if (r == hipSuccess && t) dead = true;
        return r;
    }

    // The plan of this step's sort of n elements.  Returns false when the stand-by kernel reported a barrier time-out.
    bool plan(uint32_t n, SortPlan* out) {
        uint32_t S = 0;
        while ((1u << S) < n) ++S;
        out->fuse_stage = fixed_stage; out->fallback = (force_single && fixed_stage > 0) ? 1 : 0; out->feedback = nullptr; out->seq = 0;
        if (dead) return false;
        if (!enabled || !fb || S < 15) return true;
        const volatile uint32_t* f = fb;
        const uint32_t s = f[0];
        if (s != seen) {
            if (f[4]) { dead = true; return false; }
            observe(s, (int)f[1], f[2] != FS_SORT_NO_PLAN, (int)f[3], S);
        }
        out->fuse_stage = stage ? stage : first_stage(S);
        out->fallback = single_standby() ? 1 : 0;
        out->feedback = fb;
        out->seq = ++seq;
        out->inject_timeout = inject_timeout ? 1 : 0;
        return true;
    }
};

}  // namespace fsd
