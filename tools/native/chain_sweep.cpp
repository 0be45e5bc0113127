// Torch-free tile-policy sweep for the two-stream System-2 prefill (QwenVLEngine.split_prefill): the GEMM chains of the decoder layers
// (28 x qkv, o + residual, gate|up, down + residual) and of the vision blocks (32 x qkv, proj + residual, gate|up, down + residual) are
// issued for the two half micro-batches on two streams, exactly the launches the engine makes minus attention / norm / rope, and timed
// under per-GEMM tile policies (force_cfg / group_m of ina_gemm_args). Every tile config gives bit-equal results (gemm_sweep.cpp checks
// that), so the policy is free to follow these measurements. Coordinate sweep: one GEMM of the chain changes at a time.
// Build: tools/native/build.sh; run from the repo root: tools/native/chain_sweep [lib] [llm|vit|all] [part|coord]
// part (default): the micro-batch cut into 1 .. 4 parts on as many streams x tile policies; coord: coordinate sweep of the per-GEMM tile config
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/internnav_amd.h"

#define HIP_OK(x)                                                                                      \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess) {                                                                        \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));          \
            exit(2);                                                                                   \
        }                                                                                              \
    } while (0)

typedef int (*gemm_fn)(const ina_gemm_args*, void*);
typedef const char* (*err_fn)(void);
static gemm_fn g_gemm;
static err_fn g_err;

__global__ void fill_bf16(uint16_t* p, size_t n, uint32_t seed, float scale) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t h = (uint32_t)i * 0x9E3779B9u ^ seed ^ (uint32_t)(i >> 32) * 0x85EBCA6Bu;
        h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
        const float v = ((h >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;
        uint32_t b = __float_as_uint(v);
        b += 0x7FFFu + ((b >> 16) & 1u);
        p[i] = (uint16_t)(b >> 16);
    }
}

static void* dalloc(size_t bytes, uint32_t seed, float scale) {
    void* p;
    HIP_OK(hipMalloc(&p, bytes));
    hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, 0, (uint16_t*)p, bytes / 2, seed, scale);
    HIP_OK(hipGetLastError());
    return p;
}

// one GEMM of a chain: reads activation buffer `in`, writes buffer `out` (residual GEMMs accumulate into the f32 stream in place)
struct G {
    const char* name;
    int N, K, glu, res, act;
    int in, out;                 // activation buffer ids
};
struct Policy { int cfg[4]; int gm[4]; };

struct ChainDef {
    const char* name;
    int layers;
    G g[4];
    int width[4];                // row width (elements) of the 4 activation buffers, bytes per element in esz
    int esz[4];
};

struct Inst {                    // one half micro-batch: its own activation buffers, shared weights
    int M;
    void* act[4];
};

static void launch_one(const ChainDef& c, const Inst& x, void* const W[4], int gi, const Policy& p, hipStream_t st) {
    const G& g = c.g[gi];
    ina_gemm_args a;
    memset(&a, 0, sizeof a);
    a.A = x.act[g.in]; a.W = W[gi]; a.C = x.act[g.out];
    a.R = g.res ? x.act[g.out] : nullptr;
    a.M = x.M; a.N = g.N; a.K = g.K;
    a.lda = g.K; a.ldw = g.K;
    a.ldc = g.glu ? g.N / 2 : g.N; a.ldr = a.ldc;
    a.act = g.act;
    a.out_dtype = g.res ? INA_F32 : INA_BF16;
    a.res_dtype = INA_F32;
    a.glu = g.glu; a.rowscale_div = 1; a.batch = 1;
    a.force_cfg = p.cfg[gi]; a.group_m = p.gm[gi];
    if (g_gemm(&a, (void*)st) != 0) { fprintf(stderr, "ina_gemm_bf16 failed (%s cfg %d): %s\n", g.name, p.cfg[gi], g_err()); exit(3); }
}

static Inst make_inst(const ChainDef& c, int M, uint32_t seed) {
    Inst x;
    x.M = M;
    for (int k = 0; k < 4; ++k) x.act[k] = dalloc((size_t)M * c.width[k] * c.esz[k], seed + k, 1.0f);
    return x;
}
static void free_inst(Inst& x) { for (void* p : x.act) HIP_OK(hipFree(p)); }

static hipStream_t s1, s2;
static hipEvent_t e0, e1, ej;

// halves on two streams, issue order interleaved layer by layer; returns ms (min and median of reps)
static void time_two_streams(const ChainDef& c, const Inst& a, const Inst& b, void* const W[4], const Policy& p, int reps, double& mn, double& med) {
    std::vector<double> t;
    for (int r = 0; r < reps; ++r) {
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipEventRecord(e0, s1));
        HIP_OK(hipStreamWaitEvent(s2, e0, 0));
        for (int l = 0; l < c.layers; ++l) {
            for (int gi = 0; gi < 4; ++gi) launch_one(c, a, W, gi, p, s1);
            for (int gi = 0; gi < 4; ++gi) launch_one(c, b, W, gi, p, s2);
        }
        HIP_OK(hipEventRecord(ej, s2));
        HIP_OK(hipStreamWaitEvent(s1, ej, 0));
        HIP_OK(hipEventRecord(e1, s1));
        HIP_OK(hipEventSynchronize(e1));
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    mn = t[0]; med = t[t.size() / 2];
}

// the micro-batch cut into n parts on n streams (issue order interleaved layer by layer, fork / join on stream 0)
static hipStream_t sN[4];
static hipEvent_t eN[4];
static void time_n_streams(const ChainDef& c, const Inst* parts, int n, void* const W[4], const Policy& p, int reps, double& mn, double& med) {
    std::vector<double> t;
    for (int r = 0; r < reps; ++r) {
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipEventRecord(e0, sN[0]));
        for (int k = 1; k < n; ++k) HIP_OK(hipStreamWaitEvent(sN[k], e0, 0));
        for (int l = 0; l < c.layers; ++l)
            for (int k = 0; k < n; ++k)
                for (int gi = 0; gi < 4; ++gi) launch_one(c, parts[k], W, gi, p, sN[k]);
        for (int k = 1; k < n; ++k) {
            HIP_OK(hipEventRecord(eN[k], sN[k]));
            HIP_OK(hipStreamWaitEvent(sN[0], eN[k], 0));
        }
        HIP_OK(hipEventRecord(e1, sN[0]));
        HIP_OK(hipEventSynchronize(e1));
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    mn = t[0]; med = t[t.size() / 2];
}

// partitions of a micro-batch of `envs` sequences (rows_per_env rows each) over 1 .. 4 streams x tile policies
static void partitions(const ChainDef& c, int rows_per_env, void* const W[4], int reps) {
    double flop_row = 0;
    for (int gi = 0; gi < 4; ++gi) flop_row += 2.0 * c.g[gi].N * c.g[gi].K;
    flop_row *= c.layers;
    struct Part { int n; int envs[4]; };
    const Part parts7[] = {{1, {7}}, {2, {4, 3}}, {3, {3, 2, 2}}, {4, {2, 2, 2, 1}}};
    const Part parts6[] = {{1, {6}}, {2, {3, 3}}, {3, {2, 2, 2}}, {4, {2, 2, 1, 1}}};
    const Policy pols[] = {{{0, 0, 0, 0}, {0, 0, 0, 0}}, {{-1, -1, -1, -1}, {0, 0, 0, 0}}, {{18, 18, 18, 18}, {0, 0, 0, 0}}, {{21, 21, 21, 21}, {0, 0, 0, 0}}};
    const char* pname[] = {"auto (0)", "auto, shared tail (-1)", "all cfg18", "all cfg21"};
    for (int set = 0; set < 2; ++set) {
        const Part* P = set == 0 ? parts7 : parts6;
        for (int k = 0; k < 4; ++k) {
            Inst inst[4];
            int rows = 0;
            char desc[64] = "", *d = desc;
            for (int q = 0; q < P[k].n; ++q) {
                inst[q] = make_inst(c, P[k].envs[q] * rows_per_env, 100 * q + 7);
                rows += inst[q].M;
                d += snprintf(d, desc + sizeof desc - d, q ? "+%d" : "%d", P[k].envs[q]);
            }
            HIP_OK(hipDeviceSynchronize());
            for (int pi = 0; pi < 4; ++pi) {
                if (P[k].n == 1 && pi == 1) continue;          // one stream: nothing shares the tail
                double mn, med;
                time_n_streams(c, inst, P[k].n, W, pols[pi], reps, mn, med);
                printf("%-4s envs %-8s on %d stream(s)  %-24s min %8.3f ms  median %8.3f ms  %7.1f TF/s\n", c.name, desc, P[k].n, pname[pi], mn, med,
                       flop_row * rows / (mn * 1e-3) * 1e-12);
                fflush(stdout);
            }
            for (int q = 0; q < P[k].n; ++q) free_inst(inst[q]);
        }
    }
}

static void time_one_stream(const ChainDef& c, const Inst& a, void* const W[4], const Policy& p, int reps, double& mn, double& med) {
    std::vector<double> t;
    for (int r = 0; r < reps; ++r) {
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipEventRecord(e0, s1));
        for (int l = 0; l < c.layers; ++l)
            for (int gi = 0; gi < 4; ++gi) launch_one(c, a, W, gi, p, s1);
        HIP_OK(hipEventRecord(e1, s1));
        HIP_OK(hipEventSynchronize(e1));
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    mn = t[0]; med = t[t.size() / 2];
}

static void sweep(const ChainDef& c, const int (*halves)[2], int n_halves, const int* joint, void* const W[4], int reps) {
    double flop_row = 0;
    for (int gi = 0; gi < 4; ++gi) flop_row += 2.0 * c.g[gi].N * c.g[gi].K;
    flop_row *= c.layers;
    const int alt_cfg[] = {21, 14, 11, 17, 23};
    const int alt_gm[] = {1, 4, 16};
    for (int h = 0; h < n_halves; ++h) {
        Inst a = make_inst(c, halves[h][0], 100), b = make_inst(c, halves[h][1], 200), j = make_inst(c, joint[h], 300);
        HIP_OK(hipDeviceSynchronize());
        const int rows = halves[h][0] + halves[h][1];
        auto report = [&](const char* what, const char* mode, double mn, double med, int r) {
            printf("%-4s rows %5d  %-44s %-12s min %8.3f ms  median %8.3f ms  %7.1f TF/s\n", c.name, r, what, mode, mn, med, flop_row * r / (mn * 1e-3) * 1e-12);
            fflush(stdout);
        };
        double mn, med;
        const Policy autop{{0, 0, 0, 0}, {0, 0, 0, 0}}, all18{{18, 18, 18, 18}, {0, 0, 0, 0}};
        time_one_stream(c, j, W, autop, reps, mn, med);  report("auto", "joint", mn, med, joint[h]);
        time_one_stream(c, j, W, all18, reps, mn, med);  report("all cfg18", "joint", mn, med, joint[h]);
        time_two_streams(c, a, b, W, autop, reps, mn, med);  report("auto", "2 streams", mn, med, rows);
        time_two_streams(c, a, b, W, all18, reps, mn, med);  report("all cfg18 (baseline of the sweep)", "2 streams", mn, med, rows);
        for (int gi = 0; gi < 4; ++gi) {
            for (int cf : alt_cfg) {
                Policy p = all18;
                p.cfg[gi] = cf;
                char what[96];
                snprintf(what, sizeof what, "%s -> cfg%d", c.g[gi].name, cf);
                time_two_streams(c, a, b, W, p, reps, mn, med);
                report(what, "2 streams", mn, med, rows);
            }
            for (int gm : alt_gm) {
                Policy p = all18;
                p.gm[gi] = gm;
                char what[96];
                snprintf(what, sizeof what, "%s -> cfg18 group_m %d", c.g[gi].name, gm);
                time_two_streams(c, a, b, W, p, reps, mn, med);
                report(what, "2 streams", mn, med, rows);
            }
        }
        time_two_streams(c, a, b, W, all18, reps, mn, med);  report("all cfg18 (again: drift check)", "2 streams", mn, med, rows);
        time_two_streams(c, a, b, W, autop, reps, mn, med);  report("auto (again)", "2 streams", mn, med, rows);
        free_inst(a); free_inst(b); free_inst(j);
    }
}

int main(int argc, char** argv) {
    const char* lib = argc > 1 ? argv[1] : "internnav_amd/libinternnav_amd.so";
    const std::string which = argc > 2 ? argv[2] : "all";
    void* h = dlopen(lib, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", lib, dlerror()); return 1; }
    g_gemm = (gemm_fn)dlsym(h, "ina_gemm_bf16");
    g_err = (err_fn)dlsym(h, "ina_last_error");
    typedef int (*chk_fn)(char*, int);
    chk_fn chk = (chk_fn)dlsym(h, "ina_device_check");
    if (!g_gemm || !g_err || !chk) { fprintf(stderr, "missing symbols\n"); return 1; }
    char arch[128] = {0};
    if (chk(arch, sizeof arch) != 0) { fprintf(stderr, "device check: %s\n", g_err()); return 1; }
    printf("# device %s; min / median over the repetitions of one chain; TF/s from the minimum\n", arch);
    HIP_OK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    HIP_OK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1)); HIP_OK(hipEventCreate(&ej));
    for (int k = 0; k < 4; ++k) { HIP_OK(hipStreamCreateWithFlags(&sN[k], hipStreamNonBlocking)); HIP_OK(hipEventCreate(&eN[k])); }
    const std::string mode = argc > 3 ? argv[3] : "part";          // part = partitions over 1 .. 4 streams, coord = coordinate sweep of the tile policy

    const int H = 3584, I = 18944, QKV = 4608, VH = 1280, VI = 3456;
    if (mode == "spec") {
        // tools/native/chain_sweep <lib> <spec file> spec: one line per measurement, `chain rowsA rowsB cfg_qkv cfg_o cfg_gateup cfg_down [reps]`
        // (chain = llm | vit | s1; rowsB = 0: one stream; two streams otherwise); `#` lines are echoed
        const int SD = 384, SF = 1024;
        ChainDef defs[3] = {
            {"llm", 28, {{"qkv", QKV, H, 0, 0, INA_ACT_NONE_C, 0, 1}, {"o+res", H, H, 0, 1, INA_ACT_NONE_C, 0, 2},
                         {"gate|up", 2 * I, H, 1, 0, INA_ACT_SILU_C, 0, 3}, {"down+res", H, I, 0, 1, INA_ACT_NONE_C, 3, 2}}, {H, QKV, H, I}, {2, 2, 4, 2}},
            {"vit", 32, {{"qkv", 3 * VH, VH, 0, 0, INA_ACT_NONE_C, 0, 1}, {"proj+res", VH, VH, 0, 1, INA_ACT_NONE_C, 0, 2},
                         {"gate|up", 2 * VI, VH, 1, 0, INA_ACT_SILU_C, 0, 3}, {"down+res", VH, VI, 0, 1, INA_ACT_NONE_C, 3, 2}}, {VH, 3 * VH, VH, VI}, {2, 2, 4, 2}},
            // the four tiled GEMMs of a NextDiT block at d = 384 (fused q|k|v|q2, attention out + f32 residual, SwiGLU gate|up, down + f32 residual), 48 blocks
            {"s1", 48, {{"qkvq2", 4 * SD, SD, 0, 0, INA_ACT_NONE_C, 0, 1}, {"out+res", SD, SD, 0, 1, INA_ACT_NONE_C, 0, 2},
                        {"gate|up", 2 * SF, SD, 1, 0, INA_ACT_SILU_C, 0, 3}, {"down+res", SD, SF, 0, 1, INA_ACT_NONE_C, 3, 2}}, {SD, 4 * SD, SD, SF}, {2, 2, 4, 2}}};
        void* Wt[3][4];
        for (int d = 0; d < 3; ++d)
            for (int gi = 0; gi < 4; ++gi) Wt[d][gi] = dalloc((size_t)defs[d].g[gi].N * defs[d].g[gi].K * 2, 40 + 4 * d + gi, 0.03f);
        FILE* f = fopen(which.c_str(), "r");
        if (!f) { fprintf(stderr, "cannot open %s\n", which.c_str()); return 1; }
        char line[512];
        Inst ia{0, {}}, ib{0, {}};
        int cur = -1;
        while (fgets(line, sizeof line, f)) {
            if (line[0] == '#' || line[0] == '\n') { if (line[0] == '#') fputs(line, stdout); continue; }
            char cn[16];
            int ra, rb, c0, c1, c2, c3, reps = 3;
            if (!strncmp(line, "mix", 3)) {
                // `mix s1_rows s1_blocks decode_rows decode_passes [reps]`: the concurrent phase of the n1_dual step reduced to its GEMMs - the d = 384 chain of the
                // System-1 call (s1_blocks x 4 tiled GEMMs over s1_rows rows) on one stream beside decode_passes weight-streaming passes of the decoder
                // (28 x 4 GEMMs at M = decode_rows <= 64) on another: each chain alone, then both together with the end time of each stream
                int sr, sb, dr, dp, prio = 0, dcfg = 0;
                if (sscanf(line, "%*s %d %d %d %d %d %d %d", &sr, &sb, &dr, &dp, &reps, &prio, &dcfg) < 4) continue;
                // optional 6th number: priority of the DECODE stream (hipStreamCreateWithPriority; -1 = high, 0 = default, 1 = low -> then System-1 is the default one)
                hipStream_t sdec = s2;
                if (prio != 0) {
                    int lo = 0, hi = 0;
                    HIP_OK(hipDeviceGetStreamPriorityRange(&lo, &hi));
                    HIP_OK(hipStreamCreateWithPriority(&sdec, hipStreamNonBlocking, prio < 0 ? hi : lo));
                    printf("# decode stream priority %d (device range: lowest %d .. highest %d)\n", prio < 0 ? hi : lo, lo, hi);
                }
                ChainDef cs = defs[2], cd = defs[0];
                cs.layers = sb;
                Inst xs = make_inst(cs, sr, 300), xd = make_inst(cd, dr, 400);
                // the decoder passes stream DISTINCT weights per layer (14.1 GB per pass, as the engine does): one shared layer would sit in the Infinity Cache
                static void* Wd[28][4];
                if (!Wd[0][0])
                    for (int l = 0; l < 28; ++l)
                        for (int gi = 0; gi < 4; ++gi) Wd[l][gi] = dalloc((size_t)cd.g[gi].N * cd.g[gi].K * 2, 900 + 4 * l + gi, 0.03f);
                const Policy au{{0, 0, 0, 0}, {0, 0, 0, 0}};
                const Policy pd{{dcfg, dcfg, dcfg, dcfg}, {0, 0, 0, 0}};      // optional 7th number: force_cfg of the decoder GEMMs (31 = split-K + epilogue kernel, 32 = fused)
                if (dcfg) printf("# decoder GEMMs forced to cfg %d\n", dcfg);
                hipEvent_t es, ed;
                HIP_OK(hipEventCreate(&es)); HIP_OK(hipEventCreate(&ed));
                auto run = [&](bool with_s1, bool with_dec, double& t_s1, double& t_dec) {
                    std::vector<double> a, b;
                    for (int r = 0; r < reps; ++r) {
                        HIP_OK(hipDeviceSynchronize());
                        HIP_OK(hipEventRecord(e0, s1));
                        HIP_OK(hipStreamWaitEvent(sdec, e0, 0));
                        // issue order: one decoder pass, then its share of the System-1 blocks (the host must not starve either stream)
                        const int per = with_dec ? (cs.layers + dp - 1) / dp : cs.layers;
                        int l1 = 0;
                        for (int pass = 0; pass < (with_dec ? dp : 1); ++pass) {
                            if (with_dec)
                                for (int l = 0; l < cd.layers; ++l)
                                    for (int gi = 0; gi < 4; ++gi) launch_one(cd, xd, Wd[l % 28], gi, pd, sdec);
                            if (with_s1)
                                for (int k = 0; k < per && l1 < cs.layers; ++k, ++l1)
                                    for (int gi = 0; gi < 4; ++gi) launch_one(cs, xs, Wt[2], gi, au, s1);
                        }
                        HIP_OK(hipEventRecord(es, s1));
                        HIP_OK(hipEventRecord(ed, sdec));
                        HIP_OK(hipEventSynchronize(es)); HIP_OK(hipEventSynchronize(ed));
                        float ms1 = 0, ms2 = 0;
                        HIP_OK(hipEventElapsedTime(&ms1, e0, es));
                        HIP_OK(hipEventElapsedTime(&ms2, e0, ed));
                        a.push_back(ms1); b.push_back(ms2);
                    }
                    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
                    t_s1 = a[a.size() / 2]; t_dec = b[b.size() / 2];
                };
                double s_alone, d_alone, s_both, d_both, dummy;
                run(true, false, s_alone, dummy);
                run(false, true, dummy, d_alone);
                run(true, true, s_both, d_both);
                printf("mix  System-1 GEMM chain %d rows x %d blocks alone %8.3f ms | decoder weight-streaming %d rows x %d passes alone %8.3f ms | together: System-1 ends %8.3f, "
                       "decode ends %8.3f ms (sum alone %8.3f, max %8.3f)\n", sr, sb, s_alone, dr, dp, d_alone, s_both, d_both, s_alone + d_alone, std::max(s_alone, d_alone));
                fflush(stdout);
                free_inst(xs); free_inst(xd);
                if (sdec != s2) HIP_OK(hipStreamDestroy(sdec));
                continue;
            }
            if (sscanf(line, "%15s %d %d %d %d %d %d %d", cn, &ra, &rb, &c0, &c1, &c2, &c3, &reps) < 7) continue;
            const int d = !strcmp(cn, "llm") ? 0 : !strcmp(cn, "vit") ? 1 : 2;
            if (d != cur || ia.M != ra || ib.M != rb) {
                if (ia.M) free_inst(ia);
                if (ib.M) free_inst(ib);
                ia = make_inst(defs[d], ra, 100);
                ib.M = 0;
                if (rb) ib = make_inst(defs[d], rb, 200);
                cur = d;
            }
            const Policy p{{c0, c1, c2, c3}, {0, 0, 0, 0}};
            double flop_row = 0;
            for (int gi = 0; gi < 4; ++gi) flop_row += 2.0 * defs[d].g[gi].N * defs[d].g[gi].K;
            flop_row *= defs[d].layers;
            double mn, med;
            if (rb) time_two_streams(defs[d], ia, ib, Wt[d], p, reps, mn, med);
            else time_one_stream(defs[d], ia, Wt[d], p, reps, mn, med);
            printf("%-4s rows %6d%s  cfg %2d %2d %2d %2d  min %8.3f ms  median %8.3f ms  %7.1f TF/s\n", cn, ra + rb, rb ? " (2 streams)" : " (1 stream) ", c0, c1, c2, c3, mn, med,
                   flop_row * (ra + rb) / (mn * 1e-3) * 1e-12);
            fflush(stdout);
        }
        printf("# done\n");
        return 0;
    }
    if (which == "all" || which == "llm") {
        // activation buffers: 0 x bf16 [M,H], 1 qkv bf16 [M,4608] (the o projection reads its first H columns' worth as a stand-in), 2 residual f32 [M,H], 3 ff bf16 [M,I]
        ChainDef c{"llm", 28,
                   {{"qkv", QKV, H, 0, 0, INA_ACT_NONE_C, 0, 1}, {"o+res", H, H, 0, 1, INA_ACT_NONE_C, 0, 2},
                    {"gate|up", 2 * I, H, 1, 0, INA_ACT_SILU_C, 0, 3}, {"down+res", H, I, 0, 1, INA_ACT_NONE_C, 3, 2}},
                   {H, QKV, H, I}, {2, 2, 4, 2}};
        void* W[4] = {dalloc((size_t)QKV * H * 2, 21, 0.03f), dalloc((size_t)H * H * 2, 22, 0.03f), dalloc((size_t)2 * I * H * 2, 23, 0.03f),
                      dalloc((size_t)H * I * 2, 24, 0.01f)};
        const int halves[][2] = {{3680, 2760}, {2760, 2760}};
        const int joint[] = {6440, 5520};
        if (mode == "coord") sweep(c, halves, 2, joint, W, 3);
        else partitions(c, 920, W, 3);
        for (void* p : W) HIP_OK(hipFree(p));
    }
    if (which == "all" || which == "vit") {
        // 0 h bf16 [Np,1280], 1 qkv bf16 [Np,3840], 2 residual f32 [Np,1280], 3 ff bf16 [Np,3456]
        ChainDef c{"vit", 32,
                   {{"qkv", 3 * VH, VH, 0, 0, INA_ACT_NONE_C, 0, 1}, {"proj+res", VH, VH, 0, 1, INA_ACT_NONE_C, 0, 2},
                    {"gate|up", 2 * VI, VH, 1, 0, INA_ACT_SILU_C, 0, 3}, {"down+res", VH, VI, 0, 1, INA_ACT_NONE_C, 3, 2}},
                   {VH, 3 * VH, VH, VI}, {2, 2, 4, 2}};
        void* W[4] = {dalloc((size_t)3 * VH * VH * 2, 31, 0.03f), dalloc((size_t)VH * VH * 2, 32, 0.03f), dalloc((size_t)2 * VI * VH * 2, 33, 0.03f),
                      dalloc((size_t)VH * VI * 2, 34, 0.01f)};
        const int halves[][2] = {{12544, 9408}, {9408, 9408}};
        const int joint[] = {21952, 18816};
        if (mode == "coord") sweep(c, halves, 2, joint, W, 5);
        else partitions(c, 3136, W, 5);
        for (void* p : W) HIP_OK(hipFree(p));
    }
    printf("# done\n");
    return 0;
}
