// ORACLE — TEST INFRASTRUCTURE ONLY (see vo_common.hpp).  PARITY UNPINNED.
// Numerical core of LocalMapper::localBA.  Citations are file:line under /root/reference;
// GTSAM 4.2 factor / manifold semantics per SURVEY App. B.2 [ext].
#include "vo_ba.hpp"
#include <map>

namespace vo {

static Mat3 skew(const Vec3& v) { return Mat3{{0, -v.v[2], v.v[1], v.v[2], 0, -v.v[0], -v.v[1], v.v[0], 0}}; }
static Mat3 m3add(const Mat3& a, const Mat3& b, double sb = 1.0) { Mat3 r; for (int i = 0; i < 9; i++) r.m[i] = a.m[i] + sb * b.m[i]; return r; }
static Mat3 m3scale(const Mat3& a, double s) { Mat3 r; for (int i = 0; i < 9; i++) r.m[i] = a.m[i] * s; return r; }

// Pose3::Logmap (GTSAM 4.2 Pose3.cpp)
void pose3_logmap(const Pose& T, double xi[6]) {
    const Vec3 w = so3_logmap(T.R);
    const double t = std::sqrt(dot(w, w));
    for (int i = 0; i < 3; i++) xi[i] = w.v[i];
    if (t < 1e-10) {
        for (int i = 0; i < 3; i++) xi[3 + i] = T.t.v[i];
        return;
    }
    const Mat3 W = skew(Vec3{{w.v[0] / t, w.v[1] / t, w.v[2] / t}});
    const double Tan = std::tan(0.5 * t);
    const Vec3 WT = mat3_vec(W, T.t);
    const Vec3 WWT = mat3_vec(W, WT);
    for (int i = 0; i < 3; i++) xi[3 + i] = T.t.v[i] - (0.5 * t) * WT.v[i] + (1 - t / (2. * Tan)) * WWT.v[i];
}

// SO3::LogmapDerivative
static Mat3 so3_logmap_derivative(const Vec3& w) {
    const double theta2 = dot(w, w);
    if (theta2 <= std::numeric_limits<double>::epsilon()) return mat3_identity();
    const double theta = std::sqrt(theta2);
    const Mat3 W = skew(w);
    const Mat3 WW = mat3_mul(W, W);
    Mat3 r = mat3_identity();
    r = m3add(r, W, 0.5);
    r = m3add(r, WW, 1 / (theta * theta) - (1 + std::cos(theta)) / (2 * theta * std::sin(theta)));
    return r;
}

// Pose3::computeQforExpmapDerivative (nearZeroThreshold 1e-5)
static Mat3 computeQ(const double xi[6]) {
    const Vec3 w{{xi[0], xi[1], xi[2]}}, v{{xi[3], xi[4], xi[5]}};
    const Mat3 V = skew(v), W = skew(w);
    const Mat3 WV = mat3_mul(W, V), VW = mat3_mul(V, W), WVW = mat3_mul(WV, W);
    const Mat3 WW = mat3_mul(W, W);
    const Mat3 WWV = mat3_mul(WW, V), VWW = mat3_mul(VW, W);
    const Mat3 WVWW = mat3_mul(WVW, W), WWVW = mat3_mul(WW, mat3_mul(V, W));
    const double phi = std::sqrt(dot(w, w));
    Mat3 t1 = m3add(m3add(WV, VW), WVW, -1.0);
    Mat3 t2 = m3add(m3add(WWV, VWW), WVW, -3.0);
    Mat3 t3 = m3add(WVWW, WWVW);
    Mat3 Q = m3scale(V, -0.5);
    if (phi > 1e-5) {
        const double s = std::sin(phi), c = std::cos(phi);
        const double phi2 = phi * phi, phi3 = phi2 * phi, phi4 = phi3 * phi, phi5 = phi4 * phi;
        Q = m3add(Q, t1, (phi - s) / phi3);
        Q = m3add(Q, t2, (1 - phi2 / 2 - c) / phi4);
        Q = m3add(Q, t3, -0.5 * ((1 - phi2 / 2 - c) / phi4 - 3 * (phi - s - phi3 / 6.) / phi5));
    } else {
        Q = m3add(Q, t1, 1. / 6.);
        Q = m3add(Q, t2, -1. / 24.);
        Q = m3add(Q, t3, 1. / 120.);
    }
    return Q;
}

void pose3_logmap_derivative(const Pose& T, double J[36]) {
    double xi[6];
    pose3_logmap(T, xi);
    const Mat3 Jw = so3_logmap_derivative(Vec3{{xi[0], xi[1], xi[2]}});
    const Mat3 Q = computeQ(xi);
    const Mat3 Q2 = m3scale(mat3_mul(mat3_mul(Jw, Q), Jw), -1.0);
    for (int i = 0; i < 36; i++) J[i] = 0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            J[i * 6 + j] = Jw.m[3 * i + j];
            J[(3 + i) * 6 + j] = Q2.m[3 * i + j];
            J[(3 + i) * 6 + 3 + j] = Jw.m[3 * i + j];
        }
}

// Pose3::AdjointMap: [R 0; [t]x R  R]
void pose3_adjoint(const Pose& T, double A[36]) {
    const Mat3 tR = mat3_mul(skew(T.t), T.R);
    for (int i = 0; i < 36; i++) A[i] = 0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            A[i * 6 + j] = T.R.m[3 * i + j];
            A[(3 + i) * 6 + j] = tR.m[3 * i + j];
            A[(3 + i) * 6 + 3 + j] = T.R.m[3 * i + j];
        }
}

namespace {

struct Fac {      // one GenericProjectionFactor
    int pair, kf, lm, fi;
    bool right;
    double z[2], is;
    double r[2], Jp[2][6], Jl[2][3];
};
struct Edge {     // one BetweenFactor<Pose3>, sigma 0.01 on all six
    int a, b, fa, fb;
    Pose measured;
    double r[6], Ja[36], Jb[36];
};

// whitened residual / Jacobians of a projection factor.  Right factors carry
// body_P_sensor = extrinsics (pure x translation by the baseline): q' = q - (b,0,0).
void evalFac(const Fac& f, const Pose& T, const Vec3& p, const Rig& rig, double r[2], double Jp[2][6], double Jl[2][3]) {
    const Vec3 d{{p.v[0] - T.t.v[0], p.v[1] - T.t.v[1], p.v[2] - T.t.v[2]}};
    const Vec3 q = mat3T_vec(T.R, d);
    if (Jp) for (int a = 0; a < 2; a++) { for (int c = 0; c < 6; c++) Jp[a][c] = 0; for (int c = 0; c < 3; c++) Jl[a][c] = 0; }
    if (q.v[2] <= 0) { r[0] = r[1] = 2.0 * rig.fx * f.is; return; }   // cheirality: constant residual, zero Jacobians
    const double x = q.v[0], y = q.v[1], z = q.v[2], iz = 1.0 / z;
    const double xx = f.right ? x - (double)rig.baseline : x;
    r[0] = (rig.fx * xx * iz + rig.cx - f.z[0]) * f.is;
    r[1] = (rig.fy * y * iz + rig.cy - f.z[1]) * f.is;
    if (!Jp) return;
    const double al[2][3] = {{rig.fx * iz, 0, -rig.fx * xx * iz * iz}, {0, rig.fy * iz, -rig.fy * y * iz * iz}};
    const double S[3][3] = {{0, -z, y}, {z, 0, -x}, {-y, x, 0}};
    for (int a = 0; a < 2; a++)
        for (int c = 0; c < 3; c++) {
            Jp[a][c] = (al[a][0] * S[0][c] + al[a][1] * S[1][c] + al[a][2] * S[2][c]) * f.is;
            Jp[a][3 + c] = -al[a][c] * f.is;
            // d q / d p = R^T
            Jl[a][c] = (al[a][0] * T.R.m[3 * c + 0] + al[a][1] * T.R.m[3 * c + 1] + al[a][2] * T.R.m[3 * c + 2]) * f.is;
        }
}

// BetweenFactor<Pose3>: r = Logmap(measured^-1 * (Ta^-1 Tb)), H1 = -Hlocal * Ad((Ta^-1 Tb)^-1), H2 = Hlocal
void evalEdge(const Edge& e, const Pose& Ta, const Pose& Tb, double r[6], double Ja[36], double Jb[36]) {
    const double w = 1.0 / 0.01;
    const Pose h = pose_compose(pose_inverse(Ta), Tb);
    const Pose d = pose_compose(pose_inverse(e.measured), h);
    pose3_logmap(d, r);
    for (int i = 0; i < 6; i++) r[i] *= w;
    if (!Ja) return;
    double Hl[36], Ad[36];
    pose3_logmap_derivative(d, Hl);
    pose3_adjoint(pose_inverse(h), Ad);
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            double s = 0;
            for (int k = 0; k < 6; k++) s += Hl[i * 6 + k] * Ad[k * 6 + j];
            Ja[i * 6 + j] = -s * w;
            Jb[i * 6 + j] = Hl[i * 6 + j] * w;
        }
}

void inv3sym(const double* H, double* Hi) {     // inverse of a symmetric 3x3 by cofactors
    const double a = H[0], b = H[1], c = H[2], d = H[4], e = H[5], f = H[8];
    const double A = d * f - e * e, B = c * e - b * f, C = b * e - c * d;
    const double det = a * A + b * B + c * C;
    const double id = 1.0 / det;
    Hi[0] = A * id; Hi[1] = B * id; Hi[2] = C * id;
    Hi[3] = B * id; Hi[4] = (a * f - c * c) * id; Hi[5] = (b * c - a * e) * id;
    Hi[6] = C * id; Hi[7] = (b * c - a * e) * id; Hi[8] = (a * d - b * b) * id;
}

}  // namespace

struct ShardSpec { int rank = 0, world = 1; double lambda = 0; std::vector<double>* S = nullptr; std::vector<double>* rhs = nullptr; double* cost = nullptr; int* nFree = nullptr; };

static void localBAPassImpl(const BAProblem& P, const std::vector<uint8_t>& active, int maxIterations,
                            std::vector<Pose>& kfPose, std::vector<Vec3>& lm, std::vector<uint8_t>& kfPresent,
                            std::vector<uint8_t>& lmPresent, LMReport& rep, BAResult* stats, const ShardSpec* shard);

void localBAPass(const BAProblem& P, const std::vector<uint8_t>& active, int maxIterations,
                 std::vector<Pose>& kfPose, std::vector<Vec3>& lm, std::vector<uint8_t>& kfPresent,
                 std::vector<uint8_t>& lmPresent, LMReport& rep, BAResult* stats) {
    localBAPassImpl(P, active, maxIterations, kfPose, lm, kfPresent, lmPresent, rep, stats, nullptr);
}

void reducedSystemShard(const BAProblem& P, int rank, int world, double lambda, std::vector<double>& S,
                        std::vector<double>& rhs, double& cost, int& nFree) {
    std::vector<uint8_t> active(P.pairs.size(), 1), kp, lp;
    std::vector<Pose> poses = P.kfPose;
    std::vector<Vec3> lm = P.lm;
    LMReport rep;
    ShardSpec sp;
    sp.rank = rank; sp.world = world; sp.lambda = lambda; sp.S = &S; sp.rhs = &rhs; sp.cost = &cost; sp.nFree = &nFree;
    localBAPassImpl(P, active, 0, poses, lm, kp, lp, rep, nullptr, &sp);
}

static void localBAPassImpl(const BAProblem& P, const std::vector<uint8_t>& active, int maxIterations,
                            std::vector<Pose>& kfPose, std::vector<Vec3>& lm, std::vector<uint8_t>& kfPresent,
                            std::vector<uint8_t>& lmPresent, LMReport& rep, BAResult* stats, const ShardSpec* shard) {
    const int K = (int)P.kfPose.size(), L = (int)P.lm.size();
    kfPresent.assign(K, 0);
    lmPresent.assign(L, 0);
    std::vector<Fac> facs;
    for (size_t p = 0; p < P.pairs.size(); p++) {
        if (!active[p]) continue;
        const BAPair& bp = P.pairs[p];
        if (bp.hasLeft || bp.hasRight) { kfPresent[bp.kf] = 1; lmPresent[bp.lm] = 1; }
        if (shard && bp.lm % shard->world != shard->rank) continue;
        for (int side = 0; side < 2; side++) {
            if (side == 0 ? !bp.hasLeft : !bp.hasRight) continue;
            Fac f{};
            f.pair = (int)p; f.kf = bp.kf; f.lm = bp.lm; f.right = side == 1;
            f.z[0] = side ? bp.uR : bp.uL; f.z[1] = side ? bp.vR : bp.vL;
            const int oct = side ? bp.octR : bp.octL;
            f.is = 1.0 / (1.0 / (double)P.InvSigmaFactor[oct]);     // sigma = 1/InvSigmaFactor (:601,623,689)
            facs.push_back(f);
            kfPresent[bp.kf] = 1;
            lmPresent[bp.lm] = 1;
        }
    }
    std::vector<int> fidx(K, -1);
    int F = 0;
    for (int k = 0; k < K; k++) if (kfPresent[k] && !P.kfFixed[k]) fidx[k] = F++;
    for (Fac& f : facs) f.fi = fidx[f.kf];
    // BetweenFactor between id-consecutive keyframes of the graph (:750-768), measurement = current relative pose
    std::vector<int> order;
    for (int k = 0; k < K; k++) if (kfPresent[k]) order.push_back(k);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return P.kfId[a] < P.kfId[b]; });
    std::vector<Edge> edges;
    for (size_t i = 0; i + 1 < order.size() && (!shard || shard->rank == 0); i++) {
        Edge e{};
        e.a = order[i]; e.b = order[i + 1]; e.fa = fidx[e.a]; e.fb = fidx[e.b];
        e.measured = pose_compose(pose_inverse(kfPose[e.a]), kfPose[e.b]);
        edges.push_back(e);
    }
    // landmark -> factor lists
    std::vector<std::vector<int>> lmFacs(L);
    for (size_t i = 0; i < facs.size(); i++) lmFacs[facs[i].lm].push_back((int)i);
    if (stats) {
        stats->nResiduals = (long long)facs.size();
        stats->nFreeKF = F;
        stats->nLandmarks = 0;
        stats->sumK2 = 0;
        for (int l = 0; l < L; l++) {
            if (!lmPresent[l]) continue;
            stats->nLandmarks++;
            std::map<int, int> ks;
            for (int fi : lmFacs[l]) if (facs[fi].fi >= 0) ks[facs[fi].fi]++;
            stats->sumK2 += (long long)ks.size() * (long long)ks.size();
        }
    }

    std::vector<Pose> trialPose = kfPose;
    std::vector<Vec3> trialLm = lm;
    std::vector<double> dP((size_t)6 * std::max(F, 1), 0.0), dL((size_t)3 * L, 0.0);
    double linErr0 = 0;

    auto totalError = [&](const std::vector<Pose>& TP, const std::vector<Vec3>& TL) {
        double e = 0;
        for (const Fac& f : facs) {
            double r[2];
            evalFac(f, TP[f.kf], TL[f.lm], P.rig, r, nullptr, nullptr);
            e += r[0] * r[0] + r[1] * r[1];
        }
        for (const Edge& ed : edges) {
            double r[6];
            evalEdge(ed, TP[ed.a], TP[ed.b], r, nullptr, nullptr);
            for (int i = 0; i < 6; i++) e += r[i] * r[i];
        }
        return 0.5 * e;
    };

    LMProblemX X;
    X.linearize = [&]() {
        linErr0 = 0;
        for (Fac& f : facs) {
            evalFac(f, kfPose[f.kf], lm[f.lm], P.rig, f.r, f.Jp, f.Jl);
            linErr0 += f.r[0] * f.r[0] + f.r[1] * f.r[1];
        }
        for (Edge& e : edges) {
            evalEdge(e, kfPose[e.a], kfPose[e.b], e.r, e.Ja, e.Jb);
            for (int i = 0; i < 6; i++) linErr0 += e.r[i] * e.r[i];
        }
        linErr0 *= 0.5;
    };
    X.solve = [&](double lambda, double& linChange) {
        const int n = 6 * F;
        std::vector<double> S((size_t)n * n, 0.0), rhs(n, 0.0);
        // pose blocks from projection factors
        for (const Fac& f : facs) {
            if (f.fi < 0) continue;
            for (int a = 0; a < 2; a++)
                for (int i = 0; i < 6; i++) {
                    rhs[6 * f.fi + i] -= f.Jp[a][i] * f.r[a];
                    for (int j = 0; j < 6; j++) S[(size_t)(6 * f.fi + i) * n + 6 * f.fi + j] += f.Jp[a][i] * f.Jp[a][j];
                }
        }
        for (const Edge& e : edges) {
            const int fa = e.fa, fb = e.fb;
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 6; j++) {
                    double aa = 0, ab = 0, bb = 0;
                    for (int k = 0; k < 6; k++) {
                        aa += e.Ja[k * 6 + i] * e.Ja[k * 6 + j];
                        ab += e.Ja[k * 6 + i] * e.Jb[k * 6 + j];
                        bb += e.Jb[k * 6 + i] * e.Jb[k * 6 + j];
                    }
                    if (fa >= 0) S[(size_t)(6 * fa + i) * n + 6 * fa + j] += aa;
                    if (fb >= 0) S[(size_t)(6 * fb + i) * n + 6 * fb + j] += bb;
                    if (fa >= 0 && fb >= 0) {
                        S[(size_t)(6 * fa + i) * n + 6 * fb + j] += ab;
                        S[(size_t)(6 * fb + j) * n + 6 * fa + i] += ab;
                    }
                }
            for (int i = 0; i < 6; i++) {
                double ga = 0, gb = 0;
                for (int k = 0; k < 6; k++) { ga += e.Ja[k * 6 + i] * e.r[k]; gb += e.Jb[k * 6 + i] * e.r[k]; }
                if (fa >= 0) rhs[6 * fa + i] -= ga;
                if (fb >= 0) rhs[6 * fb + i] -= gb;
            }
        }
        if (!shard || shard->rank == 0) for (int i = 0; i < n; i++) S[(size_t)i * n + i] += lambda;
        // landmark elimination (Schur complement)
        struct LmBlk { double Hi[9], bl[3]; std::vector<int> ks; std::vector<double> W; };
        std::vector<LmBlk> blk(L);
        for (int l = 0; l < L; l++) {
            if (!lmPresent[l] || (shard && l % shard->world != shard->rank)) continue;
            LmBlk& B = blk[l];
            double Hll[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            B.bl[0] = B.bl[1] = B.bl[2] = 0;
            std::map<int, int> kslot;
            for (int fi : lmFacs[l]) {
                const Fac& f = facs[fi];
                for (int a = 0; a < 2; a++)
                    for (int i = 0; i < 3; i++) {
                        B.bl[i] -= f.Jl[a][i] * f.r[a];
                        for (int j = 0; j < 3; j++) Hll[3 * i + j] += f.Jl[a][i] * f.Jl[a][j];
                    }
                if (f.fi >= 0) {
                    auto it = kslot.find(f.fi);
                    int s;
                    if (it == kslot.end()) { s = (int)B.ks.size(); kslot[f.fi] = s; B.ks.push_back(f.fi); B.W.resize(B.W.size() + 18, 0.0); }
                    else s = it->second;
                    for (int a = 0; a < 2; a++)
                        for (int i = 0; i < 6; i++)
                            for (int j = 0; j < 3; j++) B.W[(size_t)s * 18 + i * 3 + j] += f.Jp[a][i] * f.Jl[a][j];
                }
            }
            for (int i = 0; i < 3; i++) Hll[4 * i] += lambda;
            inv3sym(Hll, B.Hi);
            for (size_t s1 = 0; s1 < B.ks.size(); s1++) {
                double WH[18];   // W_k1 * Hll^-1 (6x3)
                for (int i = 0; i < 6; i++)
                    for (int j = 0; j < 3; j++) {
                        double s = 0;
                        for (int k = 0; k < 3; k++) s += B.W[s1 * 18 + i * 3 + k] * B.Hi[3 * k + j];
                        WH[i * 3 + j] = s;
                    }
                const int k1 = B.ks[s1];
                for (int i = 0; i < 6; i++) {
                    double s = 0;
                    for (int k = 0; k < 3; k++) s += WH[i * 3 + k] * B.bl[k];
                    rhs[6 * k1 + i] -= s;
                }
                for (size_t s2 = 0; s2 < B.ks.size(); s2++) {
                    const int k2 = B.ks[s2];
                    for (int i = 0; i < 6; i++)
                        for (int j = 0; j < 6; j++) {
                            double s = 0;
                            for (int k = 0; k < 3; k++) s += WH[i * 3 + k] * B.W[s2 * 18 + j * 3 + k];
                            S[(size_t)(6 * k1 + i) * n + 6 * k2 + j] -= s;
                        }
                }
            }
        }
        if (shard) { *shard->S = S; *shard->rhs = rhs; return false; }
        std::vector<double> sol = rhs;
        if (n > 0 && !chol_solve(S, sol, n)) return false;
        for (int i = 0; i < n; i++) dP[i] = sol[i];
        for (int l = 0; l < L; l++) {
            dL[3 * l] = dL[3 * l + 1] = dL[3 * l + 2] = 0;
            if (!lmPresent[l]) continue;
            const LmBlk& B = blk[l];
            double t[3] = {B.bl[0], B.bl[1], B.bl[2]};
            for (size_t s1 = 0; s1 < B.ks.size(); s1++)
                for (int j = 0; j < 3; j++) {
                    double s = 0;
                    for (int i = 0; i < 6; i++) s += B.W[s1 * 18 + i * 3 + j] * dP[6 * B.ks[s1] + i];
                    t[j] -= s;
                }
            for (int i = 0; i < 3; i++) dL[3 * l + i] = B.Hi[3 * i] * t[0] + B.Hi[3 * i + 1] * t[1] + B.Hi[3 * i + 2] * t[2];
        }
        // linear.error(delta) on the undamped system
        double lin = 0;
        for (const Fac& f : facs)
            for (int a = 0; a < 2; a++) {
                double v = f.r[a];
                if (f.fi >= 0) for (int i = 0; i < 6; i++) v += f.Jp[a][i] * dP[6 * f.fi + i];
                for (int i = 0; i < 3; i++) v += f.Jl[a][i] * dL[3 * f.lm + i];
                lin += v * v;
            }
        for (const Edge& e : edges)
            for (int k = 0; k < 6; k++) {
                double v = e.r[k];
                for (int i = 0; i < 6; i++) {
                    if (e.fa >= 0) v += e.Ja[k * 6 + i] * dP[6 * e.fa + i];
                    if (e.fb >= 0) v += e.Jb[k * 6 + i] * dP[6 * e.fb + i];
                }
                lin += v * v;
            }
        linChange = linErr0 - 0.5 * lin;
        // trial values
        for (int k = 0; k < K; k++) trialPose[k] = fidx[k] >= 0 ? pose_retract(kfPose[k], &dP[6 * fidx[k]]) : kfPose[k];
        for (int l = 0; l < L; l++) for (int i = 0; i < 3; i++) trialLm[l].v[i] = lm[l].v[i] + dL[3 * l + i];
        return true;
    };
    X.error = [&](bool atDelta) { return atDelta ? totalError(trialPose, trialLm) : totalError(kfPose, lm); };
    X.commit = [&]() { kfPose = trialPose; lm = trialLm; };
    if (shard) {
        X.linearize();
        double dummy = 0;
        X.solve(shard->lambda, dummy);
        *shard->cost = linErr0;
        *shard->nFree = F;
        return;
    }
    LMParams prm;
    prm.maxIterations = maxIterations;     // 5 on the first pass, 10 on the second (:772-777)
    prm.relativeErrorTol = 1e-5;
    prm.absoluteErrorTol = 1e-5;
    levenbergMarquardtX(X, prm, rep);
}

// chi2 re-check with the optimised values (src/OptimizationBA.cpp:787-871, checkOutlier(R) :393-424)
void chi2Check(const BAProblem& P, const std::vector<Pose>& kfPose, const std::vector<Vec3>& lm,
               const std::vector<uint8_t>& kfPresent, const std::vector<uint8_t>& lmPresent,
               std::vector<uint8_t>& pairWrong) {
    pairWrong.assign(P.pairs.size(), 0);
    const double b = (double)P.rig.baseline;
    auto outlier = [&](const Vec3& pc, float ou, float ov, int oct, bool right) {
        const double x = right ? pc.v[0] - b : pc.v[0], y = pc.v[1], z = pc.v[2];
        if (z <= 0) return true;
        const double px = P.rig.fx * x + P.rig.cx * z, py = P.rig.fy * y + P.rig.cy * z;    // K * posC
        const double eu = (double)ou - px / z, ev = (double)ov - py / z;
        const float thresh = (float)((double)7.815f * (double)P.sigmaFactor[oct]);
        return (eu * eu + ev * ev) > (double)thresh;
    };
    for (size_t p = 0; p < P.pairs.size(); p++) {
        const BAPair& bp = P.pairs[p];
        if (!P.kfLocal[bp.kf] || !kfPresent[bp.kf] || !lmPresent[bp.lm]) continue;
        if (!bp.hasLeft && !bp.hasRight) continue;
        const Pose Tcw = pose_inverse(kfPose[bp.kf]);
        Vec3 pc = mat3_vec(Tcw.R, lm[bp.lm]);
        for (int i = 0; i < 3; i++) pc.v[i] += Tcw.t.v[i];
        if (bp.hasLeft) {
            if (outlier(pc, bp.uL, bp.vL, bp.octL, false)) pairWrong[p] = 1;
            else if (bp.hasRight && outlier(pc, bp.uR, bp.vR, bp.octR, true)) pairWrong[p] = 1;
        } else {
            if (outlier(pc, bp.uR, bp.vR, bp.octR, true)) pairWrong[p] = 1;
        }
    }
}

void localBA(const BAProblem& P, BAResult& R) {
    std::vector<uint8_t> active(P.pairs.size(), 1), kfPresent, lmPresent;
    std::vector<uint8_t> wrong(P.pairs.size(), 0);
    for (int pass = 0; pass < 2; pass++) {
        for (size_t p = 0; p < P.pairs.size(); p++) active[p] = !wrong[p];
        R.kfPose = P.kfPose;       // both passes start from the map's current values (:543-559)
        R.lm = P.lm;
        localBAPass(P, active, pass == 0 ? 5 : 10, R.kfPose, R.lm, kfPresent, lmPresent, R.rep[pass], &R);
        chi2Check(P, R.kfPose, R.lm, kfPresent, lmPresent, wrong);
        if (pass == 0) R.pairWrongPass1 = wrong;
    }
    R.pairWrong = wrong;
}

void refreshDepth(float baseline, int nKf, const double* T_wc16, int nLm, const double* lm, const uint8_t* lmOutlier, int nPairs,
                  const int* pairKf, const int* pairLm, const uint8_t* pairWrong, const float* curDepth, float* depthOut,
                  uint8_t* closeOut, uint8_t* updated) {
    (void)nLm;
    std::vector<Pose> Tcw(nKf);
    for (int k = 0; k < nKf; k++) {
        Pose T;
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T.R.m[3 * r + c] = T_wc16[16 * k + 4 * r + c]; T.t.v[r] = T_wc16[16 * k + 4 * r + 3]; }
        Tcw[k] = pose_inverse(T);
    }
    const float closeTh = baseline * 40;
    for (int p = 0; p < nPairs; p++) {
        depthOut[p] = 0.f; closeOut[p] = 0; updated[p] = 0;
        const int l = pairLm[p];
        if (pairWrong[p] || lmOutlier[l] || curDepth[p] <= 0) continue;
        const Pose& T = Tcw[pairKf[p]];
        const double z = T.R.m[6] * lm[3 * l] + T.R.m[7] * lm[3 * l + 1] + T.R.m[8] * lm[3 * l + 2] + T.t.v[2] * 1.0;
        depthOut[p] = (float)z;
        closeOut[p] = z <= (double)closeTh ? 1 : 0;
        updated[p] = 1;
    }
}

}  // namespace vo
