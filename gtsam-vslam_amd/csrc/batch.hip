// vslam_batch: B independent sequences ("lanes") tracked in lockstep, one kernel launch per stage for ALL lanes.
//
// Why: a frame of one sequence is a chain of ~25 small kernels, most of them one workgroup wide (SSC per level, the
// ordered claim resolution of the projection match, the pose LM, the pre-integration) - it cannot fill 256 CUs, and the
// part runs at most ~4 kernels of different streams concurrently (tools/launchrate.hip), so independent sessions on
// their own streams stop scaling at about two sessions' worth.  Here every stage kernel is launched once with the
// lane as a grid dimension and a per-lane argument table (common.hpp lane_entry): the one-workgroup stages become
// B-workgroup launches of the same duration.
//
// Each lane is a complete vslam_system (system.hpp: the reference's Map / KeyFrame / MapPoint bookkeeping, keyframe rule,
// local mapping) bound to images 2b / 2b+1 of ONE shared extractor and to the batch's stream.  A step is
//   host  (thread pool, per lane)  frame_begin, candidate list                      [FeatureTracker::TrackImage :1115-1122]
//   device                         images -> pyramid, extraction of 2B images, stereo match of B pairs
//   host  (thread pool, per lane)  upload block: active map points, IMU bucket      (overlaps the extraction)
//   device                         one H2D, then predict / pre-integrate / project / solve / re-chain / re-predict /
//                                  project / solve / pack for all lanes, one D2H, one synchronisation
//   host  (per lane)               retry rule on lanes whose first round failed (their own launches, rare), then
//   host  (thread pool, per lane)  frame_post: bookkeeping, keyframe insertion, hand-off to the mapping threads
// Local mapping (find new points + local BA) of a lane runs on one of the batch's few mapping threads, so the number
// of streams competing for the hardware queues stays small.
//
// A lane's results are bit-identical to the same sequence run through a single vslam_system (tests/test_gpu_batch.py):
// the batched kernels are the one-session kernels' bodies, fed per-lane argument blocks built by the same host code.
#include "system.hpp"
#include "pose_dev.hpp"
#include "imu_dev.hpp"
#include "track_dev.hpp"
#include "ba_pool.hpp"
#include "job_engine.hpp"
#include <chrono>
#include <malloc.h>

namespace vslam {
void launch_pose_batch(hipStream_t s, const PoseLane* dLanes, int B);
void launch_pose_imu_batch(hipStream_t s, const PoseLane* dLanes, int B, int ldsFactors);
void launch_imu_batch(hipStream_t s, const ImuLane* dLanes, int B);
}

namespace vslam { void ba_host_profile_print(); }
using namespace vslam;
using namespace vslam_sys;

namespace {
inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
}


// ---- the mapping engine (local_mapping = 2) ------------------------------------------------------------------------------------
// The device work of the lanes' local-mapping passes, COHORT by cohort: the jobs the lanes of a group hand over during a host
// phase of a step are collected (vslam_batch::submit_mapping) and released together when that phase ends (kick).  An engine
// thread takes EVERYTHING that has been released when it becomes free - the new-point searches of the lanes that inserted a
// keyframe, the local BAs of the lanes that did a frames earlier - and runs it as ONE batched call (vslam_find_new_points_batch /
// vslam_local_ba_batch: one launch per stage for all of them).  One engine per device, shared by the lockstep groups of that
// device: a cohort then holds the jobs of every group (twice the lanes per launch with two groups, half the launches), and
// the cohort size adapts to the load - the longer a cohort takes, the more jobs the next one finds.
// VSLAM_MAP_ENGINE_PRIVATE=1: one engine per group (round-3 first form).
struct MapEngine {
    int device = 0;
    vslam::JobEngine<vslam_system*> eng;           // queues + threads (job_engine.hpp: host-only, ThreadSanitizer-tested)
    // local-BA stage timing of the cohorts (HIP events on the engine's stream), summed since the last read
    std::atomic<int> baTimingOn{0};
    std::atomic<long long> baCohortSeq{0};
    std::mutex btMu;
    std::vector<std::pair<const char*, float>> baTimes;
    long long baTimedCohorts = 0, baTimedLanes = 0;

    void release_jobs(std::deque<vslam_system*>& np, std::deque<vslam_system*>& ba) { eng.release_jobs(np, ba); }
    // the cohort's new-point searches: one upload, one launch per kernel, one download (vslam_find_new_points_batch)
    void serve_np(std::vector<vslam_system*>& jobs) {
        std::vector<const vslam_new_points_problem*> Ps; std::vector<vslam_new_points_result*> Rs;
        for (vslam_system* s : jobs) { Ps.push_back(&s->pass.np.P); Rs.push_back(&s->pass.np.R); }
        vslam_status st;
        { SysProfScope pn(sys_prof().npNs, sys_prof().npN); st = vslam_find_new_points_batch(Ps.data(), Rs.data(), (int)jobs.size(), device); }
        char err[200];
        snprintf(err, sizeof(err), "%s", st == VSLAM_OK ? "" : vslam_last_error());
        for (vslam_system* s : jobs) s->finish_job(st, err);
    }
    // the cohort's local BAs: ONE batched call (vslam_local_ba_batch: one launch per stage for all of them)
    void serve_ba(std::vector<vslam_system*>& jobs) {
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<const vslam_ba_problem*> Ps; std::vector<vslam_ba_result*> Rs;
        for (vslam_system* s : jobs) { Ps.push_back(&s->pass.ba.P); Rs.push_back(&s->pass.ba.R); }
        // stage timing (HIP events around every launch) on every 4th cohort while it is switched on: the events are launches of
        // their own on a launch-bound path (measured: all cohorts timed costs 8 % frames/s)
        const int timing = baTimingOn.load() && (baCohortSeq.fetch_add(1) % 4) == 0;
        vslam_local_ba_set_timing(timing);
        vslam_status st;
        { SysProfScope pb(sys_prof().baNs, sys_prof().baN); st = vslam_local_ba_batch(Ps.data(), Rs.data(), (int)jobs.size(), device); }
        if (st == VSLAM_OK && timing) {
            const char* nm[32]; float ms[32]; int n = 0;
            if (vslam_local_ba_timings(nm, ms, 32, &n) == VSLAM_OK) {
                std::lock_guard<std::mutex> lk(btMu);
                baTimedCohorts++; baTimedLanes += (long long)jobs.size();
                for (int i = 0; i < n; i++) {
                    size_t j = 0;
                    for (; j < baTimes.size(); j++) if (!strcmp(baTimes[j].first, nm[i])) break;
                    if (j == baTimes.size()) baTimes.push_back({nm[i], 0.f});
                    baTimes[j].second += ms[i];
                }
            }
        }
        char err[200];
        snprintf(err, sizeof(err), "%s", st == VSLAM_OK ? "" : vslam_last_error());
        for (vslam_system* s : jobs) s->finish_job(st, err);
        const long long d = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        SysProf& p = sys_prof();
        p.mqNs += d; p.mqN++; p.mqLate += (long long)jobs.size();
        if (d > 15000000) p.mapLate++;
        long long m = p.mapMaxNs.load(); while (d > m && !p.mapMaxNs.compare_exchange_weak(m, d)) {}
    }
    void start(int nNp, int nBa) {
        eng.lanes[0].serve = [this](std::vector<vslam_system*>& j) { serve_np(j); };
        eng.lanes[1].serve = [this](std::vector<vslam_system*>& j) { serve_ba(j); };
        eng.onThreadStart = [this]() { hipSetDevice(device); };
        eng.onThreadExit = []() { vslam::thread_release(); };
        eng.start(nNp, nBa);
    }
    ~MapEngine() { eng.shutdown(); }
    // one engine per device, alive while a group of that device holds it
    static std::shared_ptr<MapEngine> acquire(int device, int nBaThreads) {
        static std::mutex gMu;
        static std::weak_ptr<MapEngine> shared[64];
        static const bool priv = getenv("VSLAM_MAP_ENGINE_PRIVATE") && atoi(getenv("VSLAM_MAP_ENGINE_PRIVATE")) != 0;
        std::lock_guard<std::mutex> lk(gMu);
        std::shared_ptr<MapEngine> e = (priv || device < 0 || device >= 64) ? nullptr : shared[device].lock();
        if (!e) {
            e = std::make_shared<MapEngine>();
            e->device = device;
            const int nNp = getenv("VSLAM_MAP_NP_THREADS") ? std::max(1, atoi(getenv("VSLAM_MAP_NP_THREADS"))) : (priv ? 2 : 1);
            e->start(nNp, nBaThreads);
            if (!priv && device >= 0 && device < 64) shared[device] = e;
        }
        return e;
    }
};

struct vslam_batch {
    int B = 0, device = 0;
    bool useImu = false;
    vslam_extractor* fe = nullptr;
    hipStream_t stream = nullptr;
    // the two pre-integrations of a step run beside the matching passes that do not depend on them (one-workgroup-per-lane
    // kernels of ~90 us each): side stream + events, as the one-session path does
    hipStream_t imuStream = nullptr; hipEvent_t evTab = nullptr, evImu0 = nullptr, evSolve0 = nullptr, evImu1 = nullptr;
    std::vector<vslam_system*> sys;
    // per-lane argument tables of one step: pinned host block + device mirror (one H2D per step)
    struct Tables {
        StereoLane* stereo; PredictLane* predict; ImuLane* imu0; ProjLane* proj0; PoseLane* pose0; ImuLane* imu1;
        RepredictLane* repredict; ProjLane* proj1; PoseLane* pose1; PackLane* pack;
    } ht{}, dt{};
    uint8_t* h_tab = nullptr; uint8_t* d_tab = nullptr; size_t tabBytes = 0;
    uint8_t* h_up = nullptr; uint8_t* d_up = nullptr; size_t upCap = 0;      // upload block (map points, IMU buckets)
    uint8_t* h_dn = nullptr; uint8_t* d_dn = nullptr; size_t dnCap = 0;      // download block (per-lane tracking state)
    double* d_res = nullptr; double* h_res = nullptr;                        // [B][64] result blocks
    uint8_t* d_zero = nullptr; size_t zeroCap = 0;                           // MapPoint::GetIsOutlier of uploaded points: all 0
    BaPool pool;                                                             // host phases
    // mapping engine (shared by the groups of this device); the jobs of the current host phase wait here for kick()
    std::shared_ptr<MapEngine> eng;
    std::deque<vslam_system*> npPend, baPend;
    std::mutex pendMu;
    // per-step scratch
    struct LaneStep { bool on = false, first = false, failed = false; vslam_status st = VSLAM_OK; char err[200] = ""; size_t upOff = 0, dnOff = 0, keyOff = 0; int N = 0, nL = 0, nR = 0; bool wantKeys = false; void* keySlot = nullptr; };
    std::vector<LaneStep> ls;
    std::vector<const uint8_t*> imgPtrs;
    StageTimer timer;
    // host-side phase times of the last step (seconds): pre, extract enqueue, fill, tables + enqueue, wait, finish, post
    double phase[8] = {0}, phaseSum[8] = {0}, subSum[6] = {0};      // subSum: begin a / serve / b, post a / serve / b
    long long nSteps = 0;

    vslam_status init(const vslam_system_config* cfgs, int n, int hostThreads, int mapThreads);
    void release();
    vslam_status ensure_up(size_t bytes);
    vslam_status ensure_dn(size_t bytes);
    vslam_status serve_requests();
    std::vector<uint8_t> rqDescs; std::vector<int> rqStart, rqBest;
    vslam_status step(const uint8_t* const* L, const uint8_t* const* R, int stride, bool onDevice, const int* frames,
                      const vslam_imu_bucket* imu, const uint8_t* mask, double* T_wc_out, vslam_frame_report* reps,
                      const uint8_t* const* nextL = nullptr, const uint8_t* const* nextR = nullptr, const uint8_t* nextMask = nullptr);
    // images of the NEXT step whose extraction was enqueued at the end of the previous one (prefetch)
    std::vector<const uint8_t*> prefetched;
    vslam_status enqueue_extraction(const uint8_t* const* L, const uint8_t* const* R, const uint8_t* mask, int stride, bool onDevice);
    static void submit_mapping(void* self, vslam_system* s) {
        vslam_batch* b = (vslam_batch*)self;
        bool now = false;
        {
            std::lock_guard<std::mutex> lk(b->pendMu);
            (s->pass.stage == MapPass::NEW_POINTS ? b->npPend : b->baPend).push_back(s);
            // mapping_delay == mapping_np_delay: the lane waits for this local BA inside the same host phase - no cohort to wait for
            now = s->pass.stage == MapPass::LOCAL_BA && s->pass.commitFrame <= s->pass.npFrame;
        }
        if (now) b->kick();
    }
    void kick() {                                  // a host phase has ended: its jobs are released to the engine
        if (!eng) return;
        std::lock_guard<std::mutex> lk(pendMu);
        if (!npPend.empty() || !baPend.empty()) eng->release_jobs(npPend, baPend);
    }
};

vslam_status vslam_batch::init(const vslam_system_config* cfgs, int n, int hostThreads, int nMapThreads) {
    if (!cfgs || n <= 0 || n > 256) { set_error("vslam_batch: 1..256 lanes"); return VSLAM_ERR_INVALID; }
    B = n; device = cfgs[0].device; useImu = cfgs[0].use_imu != 0;
    for (int b = 1; b < B; b++) {
        const vslam_system_config& c = cfgs[b];
        if (c.device != device || (c.use_imu != 0) != useImu || memcmp(&c.rig, &cfgs[0].rig, sizeof(c.rig)) || memcmp(&c.fe, &cfgs[0].fe, sizeof(c.fe)) ||
            c.local_mapping != cfgs[0].local_mapping || c.mapping_delay != cfgs[0].mapping_delay || c.mapping_np_delay != cfgs[0].mapping_np_delay) {
            set_error("vslam_batch: lanes must share device, rig, extractor parameters, IMU mode and mapping mode / delay");
            return VSLAM_ERR_INVALID;
        }
    }
    VS_HIP(hipSetDevice(device));
    {
        // Every keyframe keeps ~200 KB of key arrays on the host; by default glibc serves blocks of that size with one mmap each,
        // and a dozen host-phase threads doing that at every step queue on the process's mm lock (the copy of a key block took
        // 170 - 600 us instead of ~15).  Serve them from the (per-thread) heaps instead.  VSLAM_MALLOC_TUNE=0 leaves malloc alone.
        static std::once_flag once;
        std::call_once(once, [] {
            const char* e = getenv("VSLAM_MALLOC_TUNE");
            if (e && atoi(e) == 0) return;
            mallopt(M_MMAP_THRESHOLD, 64 << 20);
            mallopt(M_TRIM_THRESHOLD, 512 << 20);
            mallopt(M_TOP_PAD, 64 << 20);
        });
    }
    VS_HIP(vslam::create_main_stream(&stream));
    timer.stream = stream; timer.multi = true;
    VS_HIP(vslam::create_main_stream(&imuStream));
    for (hipEvent_t* e : {&evTab, &evImu0, &evSolve0, &evImu1}) VS_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    VS_CHECK(vslam_extractor_create(&cfgs[0].fe, cfgs[0].rig.width, cfgs[0].rig.height, 2 * B, device, &fe));
    VS_CHECK(fe->enable_double_output());
    VS_HIP(hipMalloc(&d_res, (size_t)B * 64 * sizeof(double)));
    VS_HIP(vslam::memset_sync(d_res, 0, (size_t)B * 64 * sizeof(double)));
    VS_HIP(hipHostMalloc(&h_res, (size_t)B * 64 * sizeof(double), hipHostMallocDefault));
    sys.assign(B, nullptr);
    ls.assign(B, LaneStep{});
    for (int b = 0; b < B; b++) {
        vslam_system* s = new (std::nothrow) vslam_system();
        if (!s) return VSLAM_ERR_INVALID;
        sys[b] = s;
        VS_CHECK(s->init(&cfgs[b], fe, 2 * b, stream));
        vslam_matcher* m = s->fm;
        m->resExternal = true;
        m->d_res = d_res + (size_t)b * 64; m->h_res = h_res + (size_t)b * 64;
        m->d_poseIO = m->d_res; m->imuIo = m->d_res + 32; m->d_poseOut = (int*)(m->d_res + 48); m->d_trCount = (int*)(m->d_res + 52);
        m->trExternal = true;
        s->mapExec = &vslam_batch::submit_mapping; s->mapExecArg = this;
    }
    // argument tables
    size_t off = 0;
    auto place = [&](size_t elem) { const size_t o = off; off = up256(off + elem * (size_t)B); return o; };
    const size_t oStereo = place(sizeof(StereoLane)), oPredict = place(sizeof(PredictLane)), oImu0 = place(sizeof(ImuLane)),
                 oProj0 = place(sizeof(ProjLane)), oPose0 = place(sizeof(PoseLane)), oImu1 = place(sizeof(ImuLane)),
                 oRep = place(sizeof(RepredictLane)), oProj1 = place(sizeof(ProjLane)), oPose1 = place(sizeof(PoseLane)), oPack = place(sizeof(PackLane));
    tabBytes = off;
    VS_HIP(hipHostMalloc((void**)&h_tab, tabBytes, hipHostMallocDefault));
    VS_HIP(hipMalloc((void**)&d_tab, tabBytes));
    memset(h_tab, 0, tabBytes);
    auto bindT = [&](Tables& t, uint8_t* base) {
        t.stereo = (StereoLane*)(base + oStereo); t.predict = (PredictLane*)(base + oPredict); t.imu0 = (ImuLane*)(base + oImu0);
        t.proj0 = (ProjLane*)(base + oProj0); t.pose0 = (PoseLane*)(base + oPose0); t.imu1 = (ImuLane*)(base + oImu1);
        t.repredict = (RepredictLane*)(base + oRep); t.proj1 = (ProjLane*)(base + oProj1); t.pose1 = (PoseLane*)(base + oPose1);
        t.pack = (PackLane*)(base + oPack);
    };
    bindT(ht, h_tab); bindT(dt, d_tab);
    if (hostThreads < 0) hostThreads = std::min(B, 8);
    pool.onExit = []() { vslam::thread_release(); };
    if (hostThreads > 1) pool.start(hostThreads - 1);
    if (cfgs[0].local_mapping == 2) {
        // mapping_threads = the engine's threads for the cohorts' batched local BAs (a second cohort may start while one is in its
        // last rounds), + its thread(s) for the cohorts' batched new-point searches
        if (nMapThreads <= 0) nMapThreads = getenv("VSLAM_BATCH_MAP_THREADS") ? std::max(1, atoi(getenv("VSLAM_BATCH_MAP_THREADS"))) : 3;
        eng = MapEngine::acquire(device, nMapThreads);
    }
    return VSLAM_OK;
}

void vslam_batch::release() {
    if (getenv("VSLAM_BATCH_PHASES") && nSteps)
        fprintf(stderr, "vslam_batch %d lanes, %lld steps, host phases (us / step): begin %.1f | images + extract enqueue %.1f | upload block %.1f | "
                        "tables + enqueue %.1f (of which waiting for the extraction %.1f) | wait %.1f | retry %.1f | post %.1f\n", B, nSteps, 1e6 * phaseSum[0] / nSteps, 1e6 * phaseSum[1] / nSteps,
                1e6 * phaseSum[2] / nSteps, 1e6 * phaseSum[3] / nSteps, 1e6 * phaseSum[7] / nSteps, 1e6 * phaseSum[4] / nSteps, 1e6 * phaseSum[5] / nSteps, 1e6 * phaseSum[6] / nSteps);
    if (getenv("VSLAM_BATCH_PHASES") && nSteps)
        fprintf(stderr, "  begin = first half (waits for the jobs that are due + commits) %.1f + requests %.1f + second half (BA collection, changePosesLCA, candidates) %.1f | "
                        "post = first half %.1f + requests %.1f + second half (new-point collection, reports) %.1f\n", 1e6 * subSum[0] / nSteps, 1e6 * subSum[1] / nSteps,
                1e6 * subSum[2] / nSteps, 1e6 * subSum[3] / nSteps, 1e6 * subSum[4] / nSteps, 1e6 * subSum[5] / nSteps);
    if (getenv("VSLAM_BATCH_PHASES")) {
        SysProf& p = sys_prof();
        auto avg = [](std::atomic<long long>& ns, std::atomic<long long>& n) { return n.load() ? 1e-3 * (double)ns.load() / (double)n.load() : 0.0; };
        fprintf(stderr, "  per-lane host work so far (us per call x calls): changePosesLCA %.1f x %lld | keyframe insertion %.1f x %lld | calcDescriptor round trip "
                        "%.1f x %lld | frame_post %.1f x %lld\n", avg(p.lcaNs, p.lcaN), p.lcaN.load(), avg(p.kfNs, p.kfN), p.kfN.load(), avg(p.descNs, p.descN),
                p.descN.load(), avg(p.postNs, p.postN), p.postN.load());
        fprintf(stderr, "  mapping passes: %.1f us x %lld (find new points %.1f, vslam_local_ba %.1f) | frames that waited for their mapper: %.1f us x %lld\n",
                avg(p.mapNs, p.mapN), p.mapN.load(), avg(p.npNs, p.npN), avg(p.baNs, p.baN), avg(p.waitNs, p.waitN), p.waitN.load());
        fprintf(stderr, "  sections (total ms): KF new_keyframe %.1f | observations %.1f | stereo refill %.1f | descriptor request %.1f | key slot %.1f | connections %.1f | "
                        "keys from block %.1f (%lld by fetch_keys) || ba_collect %.1f | np_commit_a %.1f | ba_commit_a %.1f | np_collect %.1f\n", 1e-6 * p.sec[0], 1e-6 * p.sec[1], 1e-6 * p.sec[2],
                1e-6 * p.sec[3], 1e-6 * p.sec[4], 1e-6 * p.sec[5], 1e-6 * p.sec[6], (long long)p.sec[14].load(), 1e-6 * p.sec[7], 1e-6 * p.sec[8], 1e-6 * p.sec[9], 1e-6 * p.sec[10]);
        fprintf(stderr, "  ba_collect parts (total ms): window / landmark sweep %.1f | pairs %.1f | copies %.1f\n", 1e-6 * p.sec[11], 1e-6 * p.sec[12], 1e-6 * p.sec[13]);
        vslam::ba_host_profile_print();
        vslam_sys::keys_from_block_profile_print();
        fprintf(stderr, "  local-BA cohorts: %.1f us per cohort x %lld cohorts of %.1f lanes (above 15 ms: %lld, longest %.1f ms)\n",
                avg(p.mqNs, p.mqN), p.mqN.load(), p.mqN.load() ? (double)p.mqLate.load() / (double)p.mqN.load() : 0.0, p.mapLate.load(), 1e-6 * (double)p.mapMaxNs.load());
    }
    // sessions first (each waits for its mapping job), then the mapping threads, then the shared objects
    kick();
    for (vslam_system* s : sys) if (s) { s->release(); delete s; }
    sys.clear();
    eng.reset();                                   // (the last group of the device stops the engine's threads)
    if (stream) hipStreamSynchronize(stream);
    if (imuStream) { hipStreamSynchronize(imuStream); hipStreamDestroy(imuStream); imuStream = nullptr; }
    for (hipEvent_t e : {evTab, evImu0, evSolve0, evImu1}) if (e) hipEventDestroy(e);
    evTab = evImu0 = evSolve0 = evImu1 = nullptr;
    timer.destroy();
    if (fe) vslam_extractor_destroy(fe);
    fe = nullptr;
    hipFree(d_res); hipFree(d_tab); hipFree(d_up); hipFree(d_dn); hipFree(d_zero);
    if (h_res) hipHostFree(h_res);
    if (h_tab) hipHostFree(h_tab);
    if (h_up) hipHostFree(h_up);
    if (h_dn) hipHostFree(h_dn);
    if (stream) hipStreamDestroy(stream);
    stream = nullptr;
}

// The lanes' pending device requests of a host phase (descriptor selection after a keyframe insertion / new points / a local BA's
// write-back; the write-back's depth refresh), served together: ONE launch + one synchronisation per kind instead of one round
// trip per lane on the pool threads.
vslam_status vslam_batch::serve_requests() {
    // ---- gather: MapPoint::calcDescriptor ------------------------------------------------------------------------------------
    size_t nMp = 0, nDesc = 0;
    for (vslam_system* s : sys) if (s->descReq.pending) { nMp += s->descReq.mps.size(); nDesc += s->descReq.descs.size() / 32; }
    if (nMp) {
        rqDescs.resize(nDesc * 32); rqStart.assign(1, 0); rqBest.assign(nMp, -1);
        size_t at = 0;
        for (vslam_system* s : sys) {
            DescReq& q = s->descReq;
            if (!q.pending) continue;
            if (!q.descs.empty()) memcpy(rqDescs.data() + at * 32, q.descs.data(), q.descs.size());
            for (size_t i = 0; i < q.mps.size(); i++) rqStart.push_back((int)at + q.start[i + 1]);
            at += q.descs.size() / 32;
        }
    }
    // ---- gather: MapPoint::updatePos depth / close refresh ----------------------------------------------------------------------
    size_t nPair = 0; int nKf = 0, nLm = 0;
    for (vslam_system* s : sys) if (s->refReq.pending) { nPair += s->refReq.rk.size(); nKf += s->refReq.nKf; nLm += s->refReq.nLm; }
    std::vector<int> rk, rl; std::vector<float> cur, dep; std::vector<double> pose, lm;
    std::vector<uint8_t> zeroW, zeroO, clo, up;
    if (nPair) {
        cur.resize(nPair); dep.resize(nPair); pose.resize((size_t)nKf * 16); lm.resize((size_t)nLm * 3);
        zeroW.assign(nPair, 0); zeroO.assign(std::max(nLm, 1), 0); clo.resize(nPair); up.resize(nPair);
        rk.reserve(nPair); rl.reserve(nPair);
        size_t ap = 0; int ak = 0, al = 0;
        for (vslam_system* s : sys) {
            RefreshReq& r = s->refReq;
            if (!r.pending) continue;
            for (size_t i = 0; i < r.rk.size(); i++) { rk.push_back(r.rk[i] + ak); rl.push_back(r.rl[i] + al); cur[ap + i] = r.cur[i]; }
            memcpy(pose.data() + (size_t)ak * 16, r.rpose.data(), (size_t)r.nKf * 16 * sizeof(double));
            memcpy(lm.data() + (size_t)al * 3, r.rlm.data(), (size_t)r.nLm * 3 * sizeof(double));
            ap += r.rk.size(); ak += r.nKf; al += r.nLm;
        }
    }
    // ---- device: both kinds enqueued on this thread's (high-priority) pool stream, ONE wait - each used to be a round trip of its
    //      own that queued behind the groups' wide kernels.  No staging room yet (the arena grows at the sync): the synchronous calls.
    const bool wantDesc = nMp && nDesc, wantRef = nPair > 0;
    if (wantDesc || wantRef) {
        SysProfScope ps(sys_prof().descNs, sys_prof().descN);
        const int* bestP = nullptr;
        vslam::RefreshTicket tk{};
        vslam_status sd = wantDesc ? vslam::calc_descriptors_enqueue(rqDescs.data(), rqStart.data(), (int)nMp, device, &bestP) : VSLAM_OK;
        vslam_status sr = VSLAM_OK;
        if (wantRef && (sd == VSLAM_OK || sd == VSLAM_ERR_CAPACITY))
            sr = vslam::refresh_depth_enqueue(&sys[0]->cfg.rig, nKf, pose.data(), nLm, lm.data(), zeroO.data(), (int)nPair, rk.data(), rl.data(), zeroW.data(),
                                              cur.data(), device, &tk);
        if (sd != VSLAM_OK && sd != VSLAM_ERR_CAPACITY) return sd;
        if (sr != VSLAM_OK && sr != VSLAM_ERR_CAPACITY) return sr;
        vslam::DevPool* pool = vslam::thread_pool(device);
        if (!pool) return VSLAM_ERR_HIP;
        VS_HIP(hipStreamSynchronize(pool->stream));
        if (wantDesc && sd == VSLAM_OK) memcpy(rqBest.data(), bestP, nMp * sizeof(int));
        if (wantRef && sr == VSLAM_OK) { memcpy(dep.data(), tk.dep, nPair * sizeof(float)); memcpy(clo.data(), tk.clo, nPair); memcpy(up.data(), tk.up, nPair); }
        VS_HIP(pool->sync());          // (recycles the arena; may re-allocate it - after the copies)
        if (wantDesc && sd == VSLAM_ERR_CAPACITY) VS_CHECK(vslam_calc_descriptors(rqDescs.data(), rqStart.data(), (int)nMp, device, rqBest.data()));
        if (wantRef && sr == VSLAM_ERR_CAPACITY)
            VS_CHECK(vslam_ba_refresh_depth(&sys[0]->cfg.rig, nKf, pose.data(), nLm, lm.data(), zeroO.data(), (int)nPair, rk.data(), rl.data(), zeroW.data(),
                                            cur.data(), device, dep.data(), clo.data(), up.data()));
    }
    // ---- scatter ----------------------------------------------------------------------------------------------------------------
    if (nMp) {
        size_t m = 0;
        for (vslam_system* s : sys) {
            DescReq& q = s->descReq;
            if (!q.pending) continue;
            q.best.assign(rqBest.begin() + m, rqBest.begin() + m + q.mps.size());
            m += q.mps.size();
        }
    }
    if (nPair) {
        size_t ap = 0;
        for (vslam_system* s : sys) {
            RefreshReq& r = s->refReq;
            if (!r.pending) continue;
            const size_t n = r.rk.size();
            r.dep.assign(dep.begin() + ap, dep.begin() + ap + n); r.clo.assign(clo.begin() + ap, clo.begin() + ap + n); r.up.assign(up.begin() + ap, up.begin() + ap + n);
            ap += n;
        }
    }
    return VSLAM_OK;
}

vslam_status vslam_batch::ensure_up(size_t bytes) {
    if (bytes <= upCap) return VSLAM_OK;
    VS_HIP(hipStreamSynchronize(stream));
    if (h_up) hipHostFree(h_up);
    hipFree(d_up);
    upCap = bytes + bytes / 2;
    VS_HIP(hipHostMalloc((void**)&h_up, upCap, hipHostMallocDefault));
    VS_HIP(hipMalloc((void**)&d_up, upCap));
    return VSLAM_OK;
}
vslam_status vslam_batch::ensure_dn(size_t bytes) {
    if (bytes <= dnCap) return VSLAM_OK;
    VS_HIP(hipStreamSynchronize(stream));
    if (h_dn) hipHostFree(h_dn);
    hipFree(d_dn);
    dnCap = bytes + bytes / 2;
    // (download block: written by the copy engine, READ by the host phases - keyframe key blocks of 200 KB each: cacheable host memory;
    //  the default coherent kind read at < 0.5 GB/s here)
    VS_HIP(hipHostMalloc((void**)&h_dn, dnCap, hipHostMallocNonCoherent));
    VS_HIP(hipMalloc((void**)&d_dn, dnCap));
    return VSLAM_OK;
}

#define LANE_TRY(expr)                                                                                  \
    do {                                                                                                \
        vslam_status s_ = (expr);                                                                       \
        if (s_ != VSLAM_OK) { q.st = s_; q.failed = true; snprintf(q.err, sizeof(q.err), "%s", vslam_last_error()); return; } \
    } while (0)

// level 0 of every active lane's pair + the whole extraction, on the extractor's stream
vslam_status vslam_batch::enqueue_extraction(const uint8_t* const* L, const uint8_t* const* R, const uint8_t* mask, int stride, bool onDevice) {
    if (onDevice) {
        imgPtrs.assign((size_t)2 * B, nullptr);
        for (int b = 0; b < B; b++) if (!mask || mask[b]) { imgPtrs[2 * b] = L[b]; imgPtrs[2 * b + 1] = R[b]; }
        VS_CHECK(fe->set_images_device(imgPtrs.data(), stride));
    } else {
        for (int b = 0; b < B; b++) {
            if (mask && !mask[b]) continue;
            VS_CHECK(fe->set_image_async(2 * b, L[b], stride, false));
            VS_CHECK(fe->set_image_async(2 * b + 1, R[b], stride, false));
        }
        VS_HIP(hipStreamSynchronize(fe->stream));      // callers may reuse their buffers
    }
    return fe->run();
}

vslam_status vslam_batch::step(const uint8_t* const* L, const uint8_t* const* R, int stride, bool onDevice, const int* frames,
                               const vslam_imu_bucket* imu, const uint8_t* mask, double* T_wc_out, vslam_frame_report* reps,
                               const uint8_t* const* nextL, const uint8_t* const* nextR, const uint8_t* nextMask) {
    if (!L || !R || !frames || !T_wc_out) return VSLAM_ERR_INVALID;
    VS_HIP(hipSetDevice(device));
    vslam::thread_pool_wants_priority() = true;      // this thread's pool serves the group's requests (serve_requests)
    using clk = std::chrono::steady_clock;
    auto t0 = clk::now();
    auto lap = [&](int k) { const auto t1 = clk::now(); phase[k] = std::chrono::duration<double>(t1 - t0).count(); phaseSum[k] += phase[k]; t0 = t1; };
    auto sub = [&](int k, clk::time_point& ts) { const auto t1 = clk::now(); subSum[k] += std::chrono::duration<double>(t1 - ts).count(); ts = t1; };
    nSteps++;
    int nOn = 0;
    for (int b = 0; b < B; b++) {
        LaneStep& q = ls[b];
        q = LaneStep{};
        q.on = !mask || mask[b];
        if (!q.on) continue;
        if (!L[b] || !R[b]) { set_error("vslam_batch: lane %d has no images", b); return VSLAM_ERR_INVALID; }
        q.first = frames[b] == 0;
        nOn++;
    }
    if (!nOn) return VSLAM_OK;
    auto first_error = [&]() -> vslam_status {
        for (int b = 0; b < B; b++) if (ls[b].failed) { set_error("lane %d: %s", b, ls[b].err); return ls[b].st; }
        return VSLAM_OK;
    };

    // ---- host: frame_begin + candidate lists --------------------------------------------------------------------------------
    auto ts = clk::now();
    pool.run(B, [&](int b) {
        LaneStep& q = ls[b];
        if (!q.on) return;
        hipSetDevice(device);
        vslam_system* s = sys[b];
        s->deferDevice = true;
        const vslam_status bs = s->frame_begin_a(s->ctx, frames[b], imu ? &imu[b] : nullptr);
        s->deferDevice = false;
        LANE_TRY(bs);
    });
    VS_CHECK(first_error());
    sub(0, ts);
    VS_CHECK(serve_requests());                    // descriptor selection / depth refresh of every lane that committed a job
    sub(1, ts);
    pool.run(B, [&](int b) {
        LaneStep& q = ls[b];
        if (!q.on) return;
        hipSetDevice(device);
        vslam_system* s = sys[b];
        s->deferLca = true;
        const vslam_status bs = s->frame_begin_b1(s->ctx);
        s->deferLca = false;
        LANE_TRY(bs);
        if (!s->lcaReq.pending && !q.first) q.N = s->frame_candidates(s->ctx);
    });
    VS_CHECK(first_error());
    // changePosesLCA of the lanes whose local BA landed at this step: their KeyFrame::updatePose kernels enqueued back to back on this
    // thread's stream, ONE wait (each used to be a synchronous round trip of its own on a pool thread)
    {
        std::vector<int> lanesL;
        for (int b = 0; b < B; b++) if (ls[b].on && sys[b]->lcaReq.pending) lanesL.push_back(b);
        if (!lanesL.empty()) {
            std::vector<vslam::KfUpdTicket> tk(lanesL.size());
            std::vector<vslam_status> st(lanesL.size(), VSLAM_OK);
            for (size_t i = 0; i < lanesL.size(); i++) st[i] = vslam::kf_update_pose_enqueue(&sys[lanesL[i]]->lcaReq.P, device, &tk[i]);
            vslam::DevPool* dp = vslam::thread_pool(device);
            if (!dp) return VSLAM_ERR_HIP;
            VS_HIP(hipStreamSynchronize(dp->stream));
            for (size_t i = 0; i < lanesL.size(); i++) {
                if (st[i] != VSLAM_OK) continue;
                vslam_system::LcaReq& r = sys[lanesL[i]]->lcaReq;
                if (tk[i].drop_l && r.P.n_left) memcpy(r.dl.data(), tk[i].drop_l, r.P.n_left);
                if (tk[i].drop_r && r.P.n_right) memcpy(r.dr.data(), tk[i].drop_r, r.P.n_right);
                if (tk[i].lm_xyz && r.P.n_lm) memcpy(r.xyz.data(), tk[i].lm_xyz, (size_t)3 * r.P.n_lm * sizeof(double));
            }
            VS_HIP(dp->sync());          // (recycles the arena - after the copies)
            for (size_t i = 0; i < lanesL.size(); i++) {
                if (st[i] == VSLAM_OK) continue;
                if (st[i] != VSLAM_ERR_CAPACITY) return st[i];
                vslam_system::LcaReq& r = sys[lanesL[i]]->lcaReq;      // no staging room yet: the synchronous entry point
                double poseOut[16];
                VS_CHECK(vslam_keyframe_update_pose(&r.P, device, r.dl.data(), r.dr.data(), poseOut));
            }
            pool.run((int)lanesL.size(), [&](int i) {
                const int b = lanesL[i];
                LaneStep& q = ls[b];
                vslam_system* s = sys[b];
                LANE_TRY(s->frame_begin_b2(s->ctx));
                if (!q.first) q.N = s->frame_candidates(s->ctx);
            });
            VS_CHECK(first_error());
        }
    }
    sub(2, ts);
    lap(0);

    // ---- device: images, extraction (already in flight when the previous step prefetched exactly these images) -----------
    {
        bool hit = onDevice && prefetched.size() == (size_t)2 * B;
        for (int b = 0; b < B && hit; b++) {
            const bool on = ls[b].on;
            hit = prefetched[2 * b] == (on ? L[b] : nullptr) && prefetched[2 * b + 1] == (on ? R[b] : nullptr);
        }
        prefetched.clear();
        if (!hit) {
            if (fe->countsPending) VS_CHECK(fe->wait_counts());      // (a prefetch for other images: let it finish first)
            VS_CHECK(enqueue_extraction(L, R, mask, stride, onDevice));
        }
    }
    lap(1);

    // ---- host, under the extraction: the upload block -----------------------------------------------------------------------
    size_t upBytes = 0;
    for (int b = 0; b < B; b++) {
        LaneStep& q = ls[b];
        if (!q.on || q.first) continue;
        q.upOff = upBytes;
        const int n = std::max(q.N, 1);
        upBytes += up256((size_t)n * 24) + up256((size_t)n * 32) + up256((size_t)n * 4);
        if (useImu) upBytes += up256((size_t)(7 * std::max(imu ? imu[b].n : 0, 0) + 6) * sizeof(double));
    }
    VS_CHECK(ensure_up(std::max<size_t>(upBytes, 256)));
    {
        size_t zneed = 0;
        for (int b = 0; b < B; b++) zneed = std::max(zneed, (size_t)ls[b].N);
        if (zneed > zeroCap) {
            VS_HIP(hipStreamSynchronize(stream));
            hipFree(d_zero);
            zeroCap = up256(zneed + zneed / 2 + 1024);
            VS_HIP(hipMalloc((void**)&d_zero, zeroCap));
            VS_HIP(vslam::memset_sync(d_zero, 0, zeroCap));
        }
    }
    pool.run(B, [&](int b) {
        LaneStep& q = ls[b];
        if (!q.on || q.first) return;
        vslam_system* s = sys[b];
        vslam_matcher* m = s->fm;
        const int n = std::max(q.N, 1);
        uint8_t* h = h_up + q.upOff;
        uint8_t* d = d_up + q.upOff;
        const size_t oDesc = up256((size_t)n * 24), oMsd = oDesc + up256((size_t)n * 32), oImu = oMsd + up256((size_t)n * 4);
        s->frame_fill_upload(s->ctx, (double*)h, h + oDesc, (float*)(h + oMsd));
        m->track_bind_map((double*)d, d + oDesc, (float*)(d + oMsd), d_zero, q.N);
        if (useImu) {
            s->frame_imu_input(s->ctx);
            LANE_TRY(m->imu_stage(&s->ctx.in, 0.0, (double*)(h + oImu), (double*)(d + oImu), ht.imu0[b]));
        }
    });
    VS_CHECK(first_error());
    if (upBytes) VS_HIP(hipMemcpyAsync(d_up, h_up, upBytes, hipMemcpyHostToDevice, stream));
    lap(2);

    // ---- keys of the new frames (waits for the extraction's totals) -----------------------------------------------------------
    VS_HIP(hipStreamWaitEvent(stream, fe->evDone, 0));
    {
        const auto w0 = clk::now();
        VS_CHECK(fe->wait_counts());
        phaseSum[7] += std::chrono::duration<double>(clk::now() - w0).count();      // (part of phase 3: the extraction still running)
    }
    int maxL = 0, maxR = 0, maxN = 0, nTrack = 0, ldsFactors = 0;
    size_t dnBytes = 0;
    for (int b = 0; b < B; b++) {
        LaneStep& q = ls[b];
        if (!q.on) continue;
        vslam_matcher* m = sys[b]->fm;
        VS_CHECK(m->refresh_keys(false));
        q.nL = m->nKeys[0]; q.nR = m->nKeys[1];
        maxL = std::max(maxL, q.nL); maxR = std::max(maxR, q.nR);
        if (q.first) continue;
        nTrack++;
        maxN = std::max(maxN, std::max(q.N, 1));
        q.dnOff = dnBytes;
        dnBytes += up256((size_t)std::max(q.N, 1) * 15 + (size_t)q.nL * 4 + 16);
        // a lane whose keyframe counter allows an insertion this frame (src/FeatureTracker.cpp:1262: count >= 5) gets its
        // TrackedKeys in the same download; the rarer nStereo < 80 branch falls back to fetch_keys()
        // (its nStereo < 80 branch can insert at any frame: a lane whose previous frame was near that bound gets them too;
        //  a keyframe that still arrives without its block falls back to fetch_keys())
        q.wantKeys = sys[b]->insertKeyFrameCount + 1 >= 5 || sys[b]->lastNStereo < 110;
        if (q.wantKeys) { q.keyOff = dnBytes; dnBytes += up256(key_block_layout(q.nL, q.nR).total); }
    }
    VS_CHECK(ensure_dn(std::max<size_t>(dnBytes, 256)));

    // ---- per-lane argument tables (each lane fills its own entries: on the host-phase pool) ---------------------------------
    std::vector<int> laneLds((size_t)B, 0);
    pool.run(B, [&](int b) {
        LaneStep& q = ls[b];
        StereoLane& S = ht.stereo[b];
        if (!q.on) { S.A.nL = 0; S.A.nR = 0; }
        else LANE_TRY(sys[b]->fm->stereo_lane(S));
        if (q.on) sys[b]->fm->stereoDone = true;
        const bool tr = q.on && !q.first;
        vslam_matcher* m = sys[b]->fm;
        if (!tr) {
            // idle lanes: every kernel's lane test fails on these entries (count 0 / gate closed)
            ht.predict[b] = PredictLane{};
            ht.predict[b].count = m->d_trCount; ht.predict[b].poseIO = m->d_poseIO + 24;      // (scratch slot of the result block)
            ht.predict[b].setCount = 1;
            ht.imu0[b].n = 0; ht.imu1[b].n = 0;
            ht.proj0[b].A = ProjArgs{}; ht.proj1[b].A = ProjArgs{};
            ht.proj0[b].A.gate = ht.proj1[b].A.gate = m->d_trCount + 2; ht.proj0[b].A.gateMin = ht.proj1[b].A.gateMin = 1;     // (an int that stays 0)
            ht.pose0[b].A = PoseArgs{}; ht.pose1[b].A = PoseArgs{};
            ht.pose0[b].A.gate = ht.pose1[b].A.gate = m->d_trCount + 2; ht.pose0[b].A.gateMin = ht.pose1[b].A.gateMin = 1;
            ht.repredict[b] = RepredictLane{};
            ht.pack[b] = PackLane{};
            ht.pack[b].count = m->d_trCount;
            return;
        }
        vslam_system* s = sys[b];
        hipSetDevice(device);
        LANE_TRY(m->track_begin(s->predNPose.data(), frames[b], useImu));
        const int Nub = std::max(m->trNub, 1);
        const int* Mdev = m->d_trCount + 1;
        const int* gate = m->d_poseOut;          // inlier count of the first round
        const int minIn = vslam_matcher::TRACK_MIN_INLIERS;
        m->predict_lane(ht.predict[b], 0);
        ht.predict[b].setCount = 1;
        m->proj_lane(ht.proj0[b], Nub, m->trRad, Mdev, nullptr, 0, PROJ_STEREO);
        m->proj_lane(ht.proj1[b], Nub, 4.f, Mdev, gate, minIn, PROJ_STEREO);
        m->repredict_lane(ht.repredict[b], gate, minIn);
        if (useImu) {
            m->pose_imu_lane(ht.pose0[b], Nub, Mdev, nullptr, minIn, 0, 0);
            m->pose_imu_lane(ht.pose1[b], Nub, Mdev, gate, minIn, 1, 0);
            m->imu_lane(ht.imu1[b], true);
            laneLds[b] = std::max(ht.pose0[b].I.ldsFactors, ht.pose1[b].I.ldsFactors);
        } else {
            m->pose_lane(ht.pose0[b].A, Nub, Mdev, nullptr, minIn, 0, 0);
            m->pose_lane(ht.pose1[b].A, Nub, Mdev, gate, minIn, 1, 0);
        }
        PackLane& K = ht.pack[b];
        K.N = q.N; K.nL = q.nL; K.count = m->d_trCount; K.matches = m->d_matches; K.act = m->d_trAct; K.matchedL = m->d_matchedL;
        K.flags = m->d_flags; K.flagStride = (size_t)m->poseCap; K.visLeft = m->d_trVisL; K.out = d_dn + q.dnOff;
        K.keyOut = q.wantKeys ? d_dn + q.keyOff : nullptr; K.nR = q.nR;
        q.keySlot = (q.wantKeys && s->cfg.local_mapping) ? s->reserve_key_slot(q.nL, q.nR) : nullptr;
        K.keyOut2 = (uint8_t*)q.keySlot;
        K.kps[0] = m->d_kps[0]; K.kps[1] = m->d_kps[1]; K.desc[0] = m->d_desc[0]; K.desc[1] = m->d_desc[1];
        K.rightIdxs = m->d_rightIdxs; K.leftIdxs = m->d_leftIdxs; K.depth = m->d_depth; K.closef = m->d_close;
    });
    VS_CHECK(first_error());
    for (int b = 0; b < B; b++) ldsFactors = std::max(ldsFactors, laneLds[b]);
    VS_HIP(hipMemcpyAsync(d_tab, h_tab, tabBytes, hipMemcpyHostToDevice, stream));

    // ---- device: every stage once for all lanes --------------------------------------------------------------------------------
    int t;
    launch_stereo_batch(stream, dt.stereo, B, maxL, maxR, sys[0]->cfg.rig.height, &timer);
    // (stage timing brackets launches with events on the main stream: keep the pre-integrations there when it is on)
    static const bool sideEnv = getenv("VSLAM_BATCH_IMU_SIDE") ? atoi(getenv("VSLAM_BATCH_IMU_SIDE")) != 0 : true;
    const bool side = useImu && !timer.enabled && sideEnv;
    if (nTrack) {
        if (side) {
            VS_HIP(hipEventRecord(evTab, stream));                 // tables + upload block are on the device
            VS_HIP(hipStreamWaitEvent(imuStream, evTab, 0));
            launch_imu_batch(imuStream, dt.imu0, B);
            VS_HIP(hipEventRecord(evImu0, imuStream));
        }
        t = timer.begin("track_predict"); launch_track_predict_batch(stream, dt.predict, B); timer.end(t);
        if (useImu && !side) { t = timer.begin("imu_preintegrate"); launch_imu_batch(stream, dt.imu0, B); timer.end(t); }
        launch_proj_batch(stream, dt.proj0, B, maxN, maxL, maxR, &timer);
        if (side) VS_HIP(hipStreamWaitEvent(stream, evImu0, 0));
        t = timer.begin(useImu ? "pose_imu_lm" : "pose_lm");
        if (useImu) launch_pose_imu_batch(stream, dt.pose0, B, ldsFactors); else launch_pose_batch(stream, dt.pose0, B);
        timer.end(t);
        if (side) {
            VS_HIP(hipEventRecord(evSolve0, stream));
            VS_HIP(hipStreamWaitEvent(imuStream, evSolve0, 0));
            launch_imu_batch(imuStream, dt.imu1, B);
            VS_HIP(hipEventRecord(evImu1, imuStream));
        } else if (useImu) { t = timer.begin("imu_preintegrate"); launch_imu_batch(stream, dt.imu1, B); timer.end(t); }
        t = timer.begin("track_repredict"); launch_track_repredict_batch(stream, dt.repredict, B, maxN); timer.end(t);
        launch_proj_batch(stream, dt.proj1, B, maxN, maxL, maxR, &timer, false);
        if (side) VS_HIP(hipStreamWaitEvent(stream, evImu1, 0));
        t = timer.begin(useImu ? "pose_imu_lm" : "pose_lm");
        if (useImu) launch_pose_imu_batch(stream, dt.pose1, B, ldsFactors); else launch_pose_batch(stream, dt.pose1, B);
        timer.end(t);
        t = timer.begin("pack"); launch_track_pack_batch(stream, dt.pack, B, maxN); timer.end(t);
        VS_HIP(hipGetLastError());
        VS_HIP(hipMemcpyAsync(h_res, d_res, (size_t)B * 64 * sizeof(double), hipMemcpyDeviceToHost, stream));
        if (dnBytes) VS_HIP(hipMemcpyAsync(h_dn, d_dn, dnBytes, hipMemcpyDeviceToHost, stream));
    }
    VS_HIP(hipGetLastError());
    lap(3);
    // ---- host, under the step's kernels: the local BAs' window collection + hand-over (reads the map as frame_begin left it) ----
    pool.run(B, [&](int b) {
        LaneStep& q = ls[b];
        if (!q.on || !sys[b]->pass.collectDue) return;
        hipSetDevice(device);
        LANE_TRY(sys[b]->frame_mid());
    });
    kick();                                        // the local BAs handed over: one cohort
    VS_CHECK(first_error());
    VS_HIP(hipStreamSynchronize(stream));
    lap(4);
    // ---- prefetch: the next frames' extraction runs under this step's host phases and the next step's begin ----------------
    // (the extractor alternates between two output sets, so this frame's keys stay readable for keyframe insertion)
    if (nextL && nextR && onDevice) {
        VS_CHECK(enqueue_extraction(nextL, nextR, nextMask, stride, true));
        prefetched.assign((size_t)2 * B, nullptr);
        for (int b = 0; b < B; b++) if (!nextMask || nextMask[b]) { prefetched[2 * b] = nextL[b]; prefetched[2 * b + 1] = nextR[b]; }
    }

    // ---- host: the reference's retry rule per lane (launches only for lanes whose first round failed) ------------------------
    std::vector<SysTrackState> st(B);
    for (int b = 0; b < B; b++) {
        LaneStep& q = ls[b];
        if (!q.on || q.first) continue;
        vslam_system* s = sys[b];
        vslam_matcher* m = s->fm;
        SysFrameCtx& c = s->ctx;
        VS_CHECK(m->track_finish(c.T_cw, &c.tr, useImu ? &c.imuOut : nullptr));
        const int M = c.tr.n_active, N = q.N, nL = q.nL;
        uint8_t* p = h_dn + q.dnOff;
        SysTrackState& v = st[b];
        if (!m->trRetried) {
            v.matches = (const int*)p; v.actIdx = v.matches + 2 * (size_t)N; v.matchedL = v.actIdx + N;
            v.outl = (const uint8_t*)(v.matchedL + nL); v.inF = v.outl + N; v.visL = v.inF + N;
        } else {
            // the slice is large enough for the one-session layout (M <= N)
            VS_CHECK(m->track_fetch_state(p, M, nL, N));
            v.matches = (const int*)p; p += (size_t)M * 8;
            v.actIdx = (const int*)p; p += (size_t)M * 4;
            v.matchedL = (const int*)p; p += (size_t)nL * 4;
            v.outl = p; p += M;
            v.inF = p; p += M;
            v.visL = p;
        }
        v.nL = nL;
        v.keys = (q.wantKeys && !m->trRetried) ? h_dn + q.keyOff : nullptr; v.nR = q.nR;      // (retry rounds ran after the pack)
        v.keySlot = v.keys ? q.keySlot : nullptr;
    }
    lap(5);

    // ---- host: frame_post (first frames: the one-session path on their own stereo result) ------------------------------------
    ts = clk::now();
    pool.run(B, [&](int b) {
        LaneStep& q = ls[b];
        if (!q.on) return;
        hipSetDevice(device);
        vslam_system* s = sys[b];
        if (q.first) LANE_TRY(s->frame_first(s->ctx, T_wc_out + 16 * (size_t)b, reps ? &reps[b] : nullptr));
        else {
            s->deferDevice = true;
            const vslam_status ps = s->frame_post_a(s->ctx, st[b]);
            s->deferDevice = false;
            LANE_TRY(ps);
        }
    });
    VS_CHECK(first_error());
    sub(3, ts);
    VS_CHECK(serve_requests());                    // descriptor selection of the lanes that inserted a keyframe
    sub(4, ts);
    pool.run(B, [&](int b) {
        LaneStep& q = ls[b];
        if (!q.on || q.first) return;
        hipSetDevice(device);
        LANE_TRY(sys[b]->frame_post_b(sys[b]->ctx, T_wc_out + 16 * (size_t)b, reps ? &reps[b] : nullptr));
    });
    kick();                                        // the new-point searches of the lanes that inserted a keyframe
    VS_CHECK(first_error());
    sub(5, ts);
    lap(6);
    return VSLAM_OK;
}

extern "C" {

vslam_status vslam_batch_create(const vslam_system_config* configs, int32_t lanes, int32_t host_threads, int32_t mapping_threads,
                                vslam_batch** out) {
    if (!out) return VSLAM_ERR_INVALID;
    *out = nullptr;
    vslam_batch* b = new (std::nothrow) vslam_batch();
    if (!b) return VSLAM_ERR_INVALID;
    vslam_status s = b->init(configs, lanes, host_threads, mapping_threads);
    if (s != VSLAM_OK) { b->release(); delete b; return s; }
    *out = b;
    return VSLAM_OK;
}

void vslam_batch_destroy(vslam_batch* b) {
    if (!b) return;
    b->release();
    delete b;
}

vslam_status vslam_batch_track_stereo(vslam_batch* b, const uint8_t* const* left, const uint8_t* const* right, int32_t stride,
                                      int32_t on_device, const int32_t* frame_numbers, const vslam_imu_bucket* imu,
                                      const uint8_t* lane_mask, double* T_wc_out, vslam_frame_report* reports) {
    if (!b) return VSLAM_ERR_INVALID;
    return b->step(left, right, stride, on_device != 0, frame_numbers, imu, lane_mask, T_wc_out, reports);
}

vslam_status vslam_batch_track_stereo_prefetch(vslam_batch* b, const uint8_t* const* left, const uint8_t* const* right, int32_t stride,
                                               const int32_t* frame_numbers, const vslam_imu_bucket* imu, const uint8_t* lane_mask,
                                               double* T_wc_out, vslam_frame_report* reports, const uint8_t* const* next_left,
                                               const uint8_t* const* next_right, const uint8_t* next_mask) {
    if (!b) return VSLAM_ERR_INVALID;
    return b->step(left, right, stride, true, frame_numbers, imu, lane_mask, T_wc_out, reports, next_left, next_right, next_mask);
}

vslam_system* vslam_batch_system(vslam_batch* b, int32_t lane) {
    if (!b || lane < 0 || lane >= b->B) return nullptr;
    return b->sys[lane];
}

int32_t vslam_batch_lanes(const vslam_batch* b) { return b ? b->B : 0; }

vslam_status vslam_batch_wait_mapping(vslam_batch* b) {
    if (!b) return VSLAM_ERR_INVALID;
    b->kick();
    for (vslam_system* s : b->sys) VS_CHECK(vslam_system_wait_mapping(s));
    return VSLAM_OK;
}

vslam_status vslam_batch_set_timing(vslam_batch* b, int32_t on) {
    if (!b) return VSLAM_ERR_INVALID;
    b->timer.enabled = on != 0;
    b->fe->timer.enabled = on != 0;
    return VSLAM_OK;
}

// local-BA stage timing of the mapping engine's cohorts (HIP events on its stream): switch, and read-and-reset of the sums
vslam_status vslam_batch_set_ba_timing(vslam_batch* b, int32_t on) {
    if (!b) return VSLAM_ERR_INVALID;
    if (b->eng) b->eng->baTimingOn = on ? 1 : 0;
    return VSLAM_OK;
}
vslam_status vslam_batch_ba_timings(vslam_batch* b, const char** names, float* ms, int32_t cap, int32_t* n_out, int64_t* cohorts_out, int64_t* lanes_out) {
    if (!b || !n_out || !names || !ms) return VSLAM_ERR_INVALID;
    int n = 0;
    if (cohorts_out) *cohorts_out = 0;
    if (lanes_out) *lanes_out = 0;
    if (b->eng) {
        MapEngine& e = *b->eng;
        std::lock_guard<std::mutex> lk(e.btMu);
        for (auto& t : e.baTimes) if (n < cap) { names[n] = t.first; ms[n] = t.second; n++; }
        if (cohorts_out) *cohorts_out = e.baTimedCohorts;
        if (lanes_out) *lanes_out = e.baTimedLanes;
        e.baTimes.clear(); e.baTimedCohorts = 0; e.baTimedLanes = 0;
    }
    *n_out = n;
    return VSLAM_OK;
}

// device time per stage (batched launches: all lanes) since the last read, then the host phases of the last step
vslam_status vslam_batch_timings(vslam_batch* b, const char** names, float* ms, int32_t cap, int32_t* n_out, double* host_phases7) {
    if (!b || !n_out) return VSLAM_ERR_INVALID;
    VS_HIP(hipSetDevice(b->device));
    VS_HIP(hipStreamSynchronize(b->stream));
    VS_HIP(hipStreamSynchronize(b->fe->stream));
    const char* nm[64];
    float tv[64];
    int n = 0;
    {
        const int k = b->fe->timer.read(nm, tv, 64);
        b->fe->timer.reset();
        for (int i = 0; i < k && n < cap; i++, n++) { if (names) names[n] = nm[i]; if (ms) ms[n] = tv[i]; }
    }
    {
        const int k = b->timer.read(nm, tv, 64);
        b->timer.reset();
        for (int i = 0; i < k && n < cap; i++, n++) { if (names) names[n] = nm[i]; if (ms) ms[n] = tv[i]; }
    }
    *n_out = n;
    if (host_phases7) for (int i = 0; i < 7; i++) host_phases7[i] = b->phase[i];
    return VSLAM_OK;
}

}  // extern "C"
