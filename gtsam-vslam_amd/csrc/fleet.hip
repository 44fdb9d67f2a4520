// vslam_fleet: S independent SLAM sessions (vslam_system: one camera rig / sequence each) sharing one GPU, every session
// driven by its own HOST THREAD INSIDE THE LIBRARY - the frame driver the reference keeps in its main() loop
// (src/VIOSlam.cpp:289-316: for every frame TrackStereo[IMU]) plus, per session, the optimizer thread.  The path has no
// cross-sequence exchange ("replicas only"), so sessions are the unit that fills the chip: one 752x480
// stereo pair keeps a handful of the 256 CUs busy, S sessions on S streams keep S handfuls busy.
//
// Batched mode (vslam_fleet_create_batched): the sessions are the lanes of G lockstep groups (vslam_batch, batch.hip) - one
// kernel launch per stage for all lanes of a group, one driver thread per group, so that one group's host phases run
// under another group's kernels.  Same sequences, same per-session results as the one-thread-per-session mode.
//
// The frames of a (short) rendered sequence live in HBM (or pinned host memory); a session replays them as a PING-PONG
// (0 .. n-1, n-2 .. 0, 1 ...): a continuous camera motion of any length whose map, keyframes and local BAs are the
// tracker's own - nothing is re-seeded from ground truth.
#include "matcher.hpp"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

using namespace vslam;

struct vslam_fleet {
    struct Bucket { std::vector<double> acc, gyr, ts; };
    struct Session {
        vslam_system* sys = nullptr;
        std::thread th;
        int offset = 0;           // phase of the ping-pong this session starts at
        long long step = 0;       // frames tracked so far
        // accumulated over run() calls (reset by run)
        long long frames = 0, keyframes = 0, mappings = 0, inliers = 0, lost = 0, rounds = 0, newPoints = 0, baLandmarks = 0, baPairs = 0;
        long long baRes = 0, baFree = 0, baK2 = 0, baTrials = 0, baIters = 0, baRounds = 0, active = 0;
        int minInliers = 1 << 30;
        double maxPosErr = 0, sumSqPosErr = 0;
        double seconds = 0;
        vslam_status status = VSLAM_OK;
        char error[256] = "";
    };
    std::vector<Session> ses;
    struct Group { vslam_batch* b = nullptr; std::thread th; int first = 0, count = 0; };
    std::vector<Group> groups;        // batched mode (empty: one thread per session)
    int nFrames = 0, stride = 0, onDevice = 1;
    std::vector<const uint8_t*> left, right;
    std::vector<Bucket> fwd, bwd;
    std::vector<double> Ttrue;    // n x 16
    bool useImu = false;
    int device = 0;
    // per-kernel-group device time of session 0, sampled on every `sampleEvery`-th frame (0 = off)
    int sampleEvery = 0;
    std::vector<std::pair<const char*, float>> times;
    long long sampledFrames = 0, sampledSolves = 0, sampledBa = 0, sampledCohorts = 0;
    // job control
    std::mutex mu;
    std::condition_variable cvGo, cvDone;
    long long generation = 0;
    int jobSteps = 0, running = 0;
    bool stop = false;

    static int tri(long long k, int n) {           // ping-pong index
        if (n <= 1) return 0;
        const int period = 2 * (n - 1);
        const int m = (int)(k % period);
        return m < n ? m : period - m;
    }
    void loop(int s);
    void group_loop(int g);
    void account(Session& S, int si, long long k, int idx, const vslam_frame_report& rep, const double* T);
    void add_times(const char* const* nm, const float* ms, int n) {
        for (int i = 0; i < n; i++) {
            size_t j = 0;
            for (; j < times.size(); j++) if (!strcmp(times[j].first, nm[i])) break;
            if (j == times.size()) times.push_back({nm[i], 0.f});
            times[j].second += ms[i];
        }
    }
};

void vslam_fleet::account(Session& S, int si, long long k, int idx, const vslam_frame_report& rep, const double* T) {
    S.step++; S.frames++;
    S.keyframes += rep.keyframe_inserted; S.mappings += rep.mapping_ran; S.newPoints += rep.new_points;
    S.baLandmarks += rep.ba_landmarks; S.baPairs += rep.ba_pairs;
    if (rep.mapping_ran) {
        S.baRes += rep.ba_residuals; S.baFree += rep.ba_free_kf; S.baK2 += rep.ba_sum_k2; S.baTrials += rep.ba_trials; S.baRounds += rep.ba_rounds;
        S.baIters += rep.ba_report[0].iterations + rep.ba_report[1].iterations;
    }
    if (k > 0) {
        S.inliers += rep.n_inliers; S.rounds += rep.rounds; S.active += rep.n_active;
        S.minInliers = std::min(S.minInliers, rep.n_inliers);
        if (rep.n_inliers < 50) S.lost++;
    }
    if (!Ttrue.empty()) {
        const double* G = &Ttrue[16 * (size_t)idx];
        const double dx = T[3] - G[3], dy = T[7] - G[7], dz = T[11] - G[11];
        const double e2 = dx * dx + dy * dy + dz * dz;
        S.sumSqPosErr += e2; S.maxPosErr = std::max(S.maxPosErr, std::sqrt(e2));
    }
}

// driver thread of one lockstep group: all its sessions advance one frame per vslam_batch step
void vslam_fleet::group_loop(int gi) {
    Group& Gp = groups[gi];
    hipSetDevice(device);
    const int Bn = Gp.count;
    std::vector<const uint8_t*> L(Bn), R(Bn), nL(Bn), nR(Bn);
    std::vector<int> fr(Bn), idxs(Bn);
    std::vector<vslam_imu_bucket> bk(Bn);
    std::vector<double> T((size_t)Bn * 16);
    std::vector<vslam_frame_report> reps(Bn);
    long long seen = 0;
    for (;;) {
        int steps;
        {
            std::unique_lock<std::mutex> lk(mu);
            cvGo.wait(lk, [&] { return stop || generation != seen; });
            if (stop) break;
            seen = generation;
            steps = jobSteps;
        }
        const auto t0 = std::chrono::steady_clock::now();
        vslam_status st = VSLAM_OK;
        for (int q = 0; q < steps && st == VSLAM_OK; q++) {
            const long long k = ses[Gp.first].step;          // (lockstep: the same for every session of the group)
            for (int b = 0; b < Bn; b++) {
                Session& S = ses[Gp.first + b];
                const int idx = tri(k + S.offset, nFrames);
                const int prev = k > 0 ? tri(k - 1 + S.offset, nFrames) : idx;
                idxs[b] = idx; L[b] = left[idx]; R[b] = right[idx]; fr[b] = (int)std::min<long long>(k, 1 << 30);
                bk[b] = vslam_imu_bucket{};
                if (useImu && k > 0) {
                    const Bucket& B = idx > prev ? fwd[idx] : bwd[idx];
                    bk[b].n = (int)B.ts.size(); bk[b].acceleration = B.acc.data(); bk[b].angular_velocity = B.gyr.data(); bk[b].timestamps_ns = B.ts.data();
                }
            }
            const bool sample = gi == 0 && sampleEvery > 0 && k > 0 && (k % sampleEvery) == 0;
            // tracking stages: the sampled steps; local BAs: every pass of the group's first sessions while sampling is on (passes
            // are rare - one per ~40 frames per session - and run on the mapping threads, whenever they get to them)
            const int nBaTimed = std::min(Bn, 8);
            if (gi == 0 && sampleEvery > 0) {
                vslam_batch_set_timing(Gp.b, sample ? 1 : 0);
                vslam_batch_set_ba_timing(Gp.b, 1);                  // the mapping engine's cohorts (local_mapping = 2)
                for (int b = 0; b < nBaTimed; b++) vslam_system_set_ba_timing(ses[Gp.first + b].sys, 1);      // (local_mapping = 1: inside the step)
            }
            if (onDevice) {       // the next step's images: their extraction overlaps this step's host phases
                for (int b = 0; b < Bn; b++) { const int ni = tri(k + 1 + ses[Gp.first + b].offset, nFrames); nL[b] = left[ni]; nR[b] = right[ni]; }
                st = vslam_batch_track_stereo_prefetch(Gp.b, L.data(), R.data(), stride, fr.data(), useImu ? bk.data() : nullptr, nullptr, T.data(),
                                                       reps.data(), nL.data(), nR.data(), nullptr);
            } else
                st = vslam_batch_track_stereo(Gp.b, L.data(), R.data(), stride, onDevice, fr.data(), useImu ? bk.data() : nullptr, nullptr, T.data(), reps.data());
            if (st != VSLAM_OK) break;
            if (sample) {
                const char* nm[64]; float ms[64]; int n = 0, nba = 0;
                if (vslam_batch_timings(Gp.b, nm, ms, 64, &n, nullptr) == VSLAM_OK) add_times(nm, ms, n);
                for (int b = 0; b < nBaTimed; b++)
                    if (vslam_system_ba_timings(ses[Gp.first + b].sys, nm, ms, 64, &n, &nba) == VSLAM_OK) { add_times(nm, ms, n); sampledBa += nba; sampledCohorts += nba; }
                int64_t co = 0, la = 0;
                if (vslam_batch_ba_timings(Gp.b, nm, ms, 64, &n, &co, &la) == VSLAM_OK) { add_times(nm, ms, n); sampledBa += la; sampledCohorts += co; }
                sampledFrames += Bn;
                for (int b = 0; b < Bn; b++) sampledSolves += reps[b].rounds + 1;
            }
            for (int b = 0; b < Bn; b++) account(ses[Gp.first + b], Gp.first + b, k, idxs[b], reps[b], &T[16 * (size_t)b]);
        }
        if (st == VSLAM_OK) st = vslam_batch_wait_mapping(Gp.b);      // every local BA of these frames completes inside the run
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        for (int b = 0; b < Bn; b++) {
            Session& S = ses[Gp.first + b];
            S.seconds = sec;
            if (st != VSLAM_OK && S.status == VSLAM_OK) { S.status = st; snprintf(S.error, sizeof(S.error), "%s", vslam_last_error()); }
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            if (--running == 0) cvDone.notify_all();
        }
    }
    vslam::thread_release();
}

void vslam_fleet::loop(int si) {
    Session& S = ses[si];
    hipSetDevice(device);
    long long seen = 0;
    for (;;) {
        int steps;
        {
            std::unique_lock<std::mutex> lk(mu);
            cvGo.wait(lk, [&] { return stop || generation != seen; });
            if (stop) break;
            seen = generation;
            steps = jobSteps;
        }
        const auto t0 = std::chrono::steady_clock::now();
        for (int q = 0; q < steps && S.status == VSLAM_OK; q++) {
            const long long k = S.step;
            const int idx = tri(k + S.offset, nFrames);
            const int prev = k > 0 ? tri(k - 1 + S.offset, nFrames) : idx;
            vslam_imu_bucket b{};
            const vslam_imu_bucket* bp = nullptr;
            if (useImu && k > 0) {
                const Bucket& B = idx > prev ? fwd[idx] : bwd[idx];
                b.n = (int)B.ts.size(); b.acceleration = B.acc.data(); b.angular_velocity = B.gyr.data(); b.timestamps_ns = B.ts.data();
                bp = &b;
            }
            const bool sample = si == 0 && sampleEvery > 0 && k > 0 && (k % sampleEvery) == 0;
            if (si == 0 && sampleEvery > 0) vslam_system_set_timing(S.sys, sample ? 1 : 0);
            double T[16];
            vslam_frame_report rep{};
            const vslam_status st = vslam_system_track_stereo(S.sys, left[idx], right[idx], stride, onDevice, (int)std::min<long long>(k, 1 << 30), bp, T, &rep);
            if (st != VSLAM_OK) { S.status = st; snprintf(S.error, sizeof(S.error), "%s", vslam_last_error()); break; }
            if (sample) {
                const char* nm[64]; float ms[64]; int n = 0, nba = 0;
                if (vslam_system_timings(S.sys, nm, ms, 64, &n, &nba) == VSLAM_OK) {
                    add_times(nm, ms, n);
                    sampledFrames++; sampledSolves += rep.rounds + 1; sampledBa += nba;
                }
            }
            account(S, si, k, idx, rep, T);
        }
        if (S.status == VSLAM_OK) {               // every local BA of these frames completes inside the run
            const vslam_status st = vslam_system_wait_mapping(S.sys);
            if (st != VSLAM_OK) { S.status = st; snprintf(S.error, sizeof(S.error), "%s", vslam_last_error()); }
        }
        S.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        {
            std::lock_guard<std::mutex> lk(mu);
            if (--running == 0) cvDone.notify_all();
        }
    }
    vslam::thread_release();
}

extern "C" {

static vslam_status fleet_create(const vslam_system_config* config, int32_t n_sessions, const vslam_fleet_sequence* seq, int lanes, vslam_fleet** out) {
    if (!config || !seq || !out || n_sessions < 1 || n_sessions > 1024 || seq->n_frames < 2 || !seq->left || !seq->right || seq->stride < config->rig.width) {
        set_error("vslam_fleet_create: invalid arguments");
        return VSLAM_ERR_INVALID;
    }
    if (config->use_imu && (!seq->imu_forward || !seq->imu_backward || !seq->T_wc_true)) { set_error("vslam_fleet_create: IMU mode needs the buckets and the true poses"); return VSLAM_ERR_INVALID; }
    *out = nullptr;
    vslam_fleet* F = new (std::nothrow) vslam_fleet();
    if (!F) return VSLAM_ERR_INVALID;
    F->nFrames = seq->n_frames; F->stride = seq->stride; F->onDevice = seq->on_device; F->useImu = config->use_imu != 0; F->device = config->device;
    F->left.assign((const uint8_t* const*)seq->left, (const uint8_t* const*)seq->left + seq->n_frames);
    F->right.assign((const uint8_t* const*)seq->right, (const uint8_t* const*)seq->right + seq->n_frames);
    if (seq->T_wc_true) F->Ttrue.assign(seq->T_wc_true, seq->T_wc_true + 16 * (size_t)seq->n_frames);
    auto copyB = [&](const vslam_imu_bucket* src, std::vector<vslam_fleet::Bucket>& dst) {
        dst.resize(seq->n_frames);
        for (int i = 0; i < seq->n_frames; i++) {
            const vslam_imu_bucket& b = src[i];
            if (b.n <= 0 || !b.acceleration || !b.angular_velocity || !b.timestamps_ns) continue;
            dst[i].acc.assign(b.acceleration, b.acceleration + 3 * (size_t)b.n);
            dst[i].gyr.assign(b.angular_velocity, b.angular_velocity + 3 * (size_t)b.n);
            dst[i].ts.assign(b.timestamps_ns, b.timestamps_ns + b.n);
        }
    };
    if (F->useImu) { copyB(seq->imu_forward, F->fwd); copyB(seq->imu_backward, F->bwd); }
    F->ses.resize(n_sessions);
    vslam_status st = VSLAM_OK;
    std::vector<vslam_system_config> cfgs(n_sessions, *config);
    for (int s = 0; s < n_sessions; s++) {
        vslam_system_config& c = cfgs[s];
        // sessions start at different phases of the forward leg, each at the true pose / velocity of its first frame
        const int span = seq->start_span > 0 ? std::min(seq->start_span, seq->n_frames - 1) : seq->n_frames - 1;
        const int off = (int)(((long long)s * 5) % std::max(1, span));
        F->ses[s].offset = off;
        if (seq->T_wc_true) memcpy(c.T_wc_init, seq->T_wc_true + 16 * (size_t)off, sizeof(c.T_wc_init));
        if (seq->velocity_true) for (int k = 0; k < 3; k++) c.velocity_init[k] = seq->velocity_true[3 * (size_t)off + k];
    }
    if (lanes <= 0) {
        for (int s = 0; s < n_sessions && st == VSLAM_OK; s++) st = vslam_system_create(&cfgs[s], &F->ses[s].sys);
    } else {
        const int nG = (n_sessions + lanes - 1) / lanes;
        F->groups.resize(nG);
        // host threads per group: the groups share the machine's cores
        // (one process per GPU on a multi-GPU node: this process's share of the cores - torch.distributed.run exports LOCAL_WORLD_SIZE)
        int hw = (int)std::max(2u, std::thread::hardware_concurrency());
        if (const char* lws = getenv("LOCAL_WORLD_SIZE")) hw = std::max(2, hw / std::max(1, atoi(lws)));
        int hostThreads = std::max(1, std::min(8, hw / std::max(nG, 1) - 1));
        if (const char* e = getenv("VSLAM_FLEET_HOST_THREADS")) hostThreads = std::max(1, atoi(e));
        for (int g = 0; g < nG && st == VSLAM_OK; g++) {
            vslam_fleet::Group& G = F->groups[g];
            G.first = g * lanes; G.count = std::min(lanes, n_sessions - G.first);
            st = vslam_batch_create(&cfgs[G.first], G.count, hostThreads, 0, &G.b);
            if (st == VSLAM_OK) for (int b = 0; b < G.count; b++) F->ses[G.first + b].sys = vslam_batch_system(G.b, b);
        }
    }
    if (st != VSLAM_OK) {
        if (F->groups.empty()) { for (auto& S : F->ses) if (S.sys) vslam_system_destroy(S.sys); }
        else for (auto& G : F->groups) if (G.b) vslam_batch_destroy(G.b);
        delete F;
        return st;
    }
    if (F->groups.empty()) for (int s = 0; s < n_sessions; s++) F->ses[s].th = std::thread([F, s]() { F->loop(s); });
    else for (int g = 0; g < (int)F->groups.size(); g++) F->groups[g].th = std::thread([F, g]() { F->group_loop(g); });
    *out = F;
    return VSLAM_OK;
}

vslam_status vslam_fleet_create(const vslam_system_config* config, int32_t n_sessions, const vslam_fleet_sequence* seq, vslam_fleet** out) {
    return fleet_create(config, n_sessions, seq, 0, out);
}

vslam_status vslam_fleet_create_batched(const vslam_system_config* config, int32_t n_sessions, const vslam_fleet_sequence* seq,
                                        int32_t lanes_per_group, vslam_fleet** out) {
    if (lanes_per_group < 1) { set_error("vslam_fleet_create_batched: lanes_per_group >= 1"); return VSLAM_ERR_INVALID; }
    return fleet_create(config, n_sessions, seq, lanes_per_group, out);
}

void vslam_fleet_destroy(vslam_fleet* F) {
    if (!F) return;
    { std::lock_guard<std::mutex> lk(F->mu); F->stop = true; }
    F->cvGo.notify_all();
    for (auto& S : F->ses) if (S.th.joinable()) S.th.join();
    for (auto& G : F->groups) if (G.th.joinable()) G.th.join();
    if (F->groups.empty()) { for (auto& S : F->ses) if (S.sys) vslam_system_destroy(S.sys); }
    else for (auto& G : F->groups) if (G.b) vslam_batch_destroy(G.b);
    delete F;
}

vslam_status vslam_fleet_run(vslam_fleet* F, int32_t n_steps, vslam_fleet_report* rep) {
    if (!F || n_steps < 0) return VSLAM_ERR_INVALID;
    for (auto& S : F->ses) {
        S.frames = S.keyframes = S.mappings = S.inliers = S.lost = S.rounds = S.newPoints = S.baLandmarks = S.baPairs = 0;
        S.baRes = S.baFree = S.baK2 = S.baTrials = S.baIters = S.baRounds = S.active = 0;
        S.minInliers = 1 << 30; S.maxPosErr = 0; S.sumSqPosErr = 0; S.seconds = 0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    {
        std::unique_lock<std::mutex> lk(F->mu);
        F->jobSteps = n_steps; F->running = F->groups.empty() ? (int)F->ses.size() : (int)F->groups.size(); F->generation++;
        F->cvGo.notify_all();
        F->cvDone.wait(lk, [&] { return F->running == 0; });
    }
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    vslam_fleet_report R{};
    R.n_sessions = (int)F->ses.size(); R.seconds = el; R.min_inliers = 1 << 30;
    vslam_status st = VSLAM_OK;
    for (auto& S : F->ses) {
        if (S.status != VSLAM_OK && st == VSLAM_OK) { st = S.status; set_error("fleet session failed: %s", S.error); }
        R.frames += S.frames; R.keyframes += S.keyframes; R.mappings += S.mappings; R.sum_inliers += S.inliers; R.lost_frames += S.lost;
        R.sum_rounds += S.rounds; R.sum_active += S.active; R.new_points += S.newPoints; R.ba_landmarks += S.baLandmarks; R.ba_pairs += S.baPairs;
        R.ba_residuals += S.baRes; R.ba_free_kf += S.baFree; R.ba_sum_k2 += S.baK2; R.ba_trials += S.baTrials; R.ba_iterations += S.baIters; R.ba_rounds += S.baRounds;
        R.min_inliers = std::min<int>(R.min_inliers, S.minInliers);
        R.max_position_error = std::max(R.max_position_error, S.maxPosErr);
        R.sum_sq_position_error += S.sumSqPosErr;
        R.max_session_seconds = std::max(R.max_session_seconds, S.seconds);
    }
    if (rep) *rep = R;
    return st;
}

vslam_status vslam_fleet_set_sampling(vslam_fleet* F, int32_t every) {
    if (!F || every < 0) return VSLAM_ERR_INVALID;
    F->sampleEvery = every;
    if (!every) {
        if (F->groups.empty()) vslam_system_set_timing(F->ses[0].sys, 0);
        else {
            vslam_batch_set_timing(F->groups[0].b, 0);
            vslam_batch_set_ba_timing(F->groups[0].b, 0);
            for (int b = 0; b < std::min(F->groups[0].count, 8); b++) vslam_system_set_ba_timing(F->ses[b].sys, 0);
        }
    }
    return VSLAM_OK;
}

vslam_status vslam_fleet_timings(vslam_fleet* F, const char** names, float* ms, int32_t cap, int32_t* n_out, int64_t* counts3) {
    if (!F || !n_out) return VSLAM_ERR_INVALID;
    int n = 0;
    for (auto& t : F->times) if (n < cap) { if (names) names[n] = t.first; if (ms) ms[n] = t.second; n++; }
    *n_out = n;
    if (counts3) { counts3[0] = F->sampledFrames; counts3[1] = F->sampledSolves; counts3[2] = F->sampledBa; counts3[3] = F->sampledCohorts; }
    F->times.clear(); F->sampledFrames = F->sampledSolves = F->sampledBa = F->sampledCohorts = 0;
    return VSLAM_OK;
}

vslam_status vslam_fleet_system(vslam_fleet* F, int32_t session, vslam_system** out) {
    if (!F || !out || session < 0 || session >= (int)F->ses.size()) return VSLAM_ERR_INVALID;
    *out = F->ses[session].sys;
    return VSLAM_OK;
}

}  // extern "C"
