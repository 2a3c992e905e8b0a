// Host-side logic of libmmwgpu under AddressSanitizer + UndefinedBehaviorSanitizer (the GPU build
// cannot run sanitizers on this pool, the host code can).  The library's translation units are compiled host-only
// (hipcc --cuda-host-only -fsanitize=address,undefined: kernels become launch stubs) and linked with this driver, which
// calls every entry point that needs no device: the range-Doppler / chain / detection planners over a sweep of shapes,
// the chirp-z run splitter, argument validation and error-string plumbing, context creation on a machine without a GPU.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/mmwgpu.h"

static int fails = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) {                                                       \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++fails;                                                         \
        }                                                                    \
    } while (0)

int main() {
    std::printf("%s abi %d\n", mmw_version(), mmw_abi_version());
    CHECK(mmw_abi_version() == MMWGPU_ABI_VERSION);
    int plan[8];
    long planned = 0;
    // range-Doppler planner: every (S, C) of the shipped cfgs and a sweep around them, both precisions
    const int sizes[] = {1, 2, 3, 7, 8, 16, 30, 32, 50, 63, 64, 70, 90, 100, 115, 120, 126, 127, 128, 130, 200, 254, 256, 257, 512, 1024, 4096};
    for (int S : sizes)
        for (int C : sizes)
            for (int f64 = 0; f64 < 2; ++f64) {
                CHECK(mmw_diag_rd_plan(S, C, f64, plan) == MMW_OK);
                CHECK(plan[0] >= 0 && plan[0] <= 4);
                if (plan[0] == 2) CHECK(plan[3] * plan[4] == S && plan[5] * plan[6] == C && plan[7] > 0);
                ++planned;
            }
    CHECK(mmw_diag_rd_plan(0, 8, 0, plan) == MMW_ERR_INVALID && std::strlen(mmw_last_error()) > 0);
    CHECK(mmw_diag_rd_plan(8, 8, 0, nullptr) == MMW_ERR_INVALID);
    // chain schedule planner for 256 / 304 / 64-CU devices, plain and raw cubes, tiny and huge batches
    for (int cus : {64, 256, 304})
        for (int raw = 0; raw < 2; ++raw)
            for (int F : {1, 7, 96, 1250, 65535, 1000000})
                for (int V : {4, 8, 12, 16})
                    for (int S : {63, 64, 254, 256, 512})
                        for (int C : {50, 100, 127, 128, 256})
                            for (int flags : {0, 1, 2, 5}) {
                                CHECK(mmw_diag_chain_plan_nodev(cus, raw, F, V, S, C, 64, flags, plan) == MMW_OK);
                                CHECK(plan[1] >= 1 && plan[1] <= F && plan[3] >= 0 && plan[3] < cus);
                                CHECK(plan[4] == V || plan[4] == V - 2);
                                if (plan[6]) CHECK(plan[7] >= 2 && plan[7] <= 256);
                                ++planned;
                            }
    CHECK(mmw_diag_chain_plan_nodev(256, 0, 10, 12, 256, 128, 8, 0, plan) == MMW_ERR_INVALID);        // A < V
    // banding of the fused detection stage: every window from (0,0)/(0,0) to (9,9)/(4,4) on planes up to 4096 x 256
    for (int S : {13, 32, 63, 64, 127, 254, 256, 512, 1024, 4096})
        for (int C : {16, 32, 50, 100, 127, 128, 256})
            for (int tr : {0, 1, 4, 5, 9})
                for (int gr : {0, 2, 3, 4})
                    for (int n_az : {0, 4, 8, 9, 16, 17})
                      for (int A : {64, 128}) {
                        CHECK(mmw_diag_detect_plan(S, C, MMW_CFAR_CA, tr, tr, gr, gr > 2 ? 2 : gr, n_az, 4, A, plan) == MMW_OK);
                        const bool expect = mmw_detect_points_supported(S, C, MMW_CFAR_CA, tr, tr, gr, gr > 2 ? 2 : gr, n_az, 4, A) != 0;
                        CHECK((plan[0] != 0) == expect);
                        if (plan[0]) {
                            CHECK(plan[1] == 1 && plan[5] > 0 && plan[5] <= 160 * 1024 && plan[7] >= 8);
                            // a band and its halo rows fit the loads a workgroup keeps in flight
                            CHECK(plan[3] >= 1 && plan[3] + 2 * (tr + gr) <= plan[2]);
                        }
                        // lists of 9 to 16 antennas: only with 64 angle bins (the late argmax); longer ones never
                        if (n_az > 16 || (n_az > 8 && A != 64)) CHECK(plan[0] == 0);
                        ++planned;
                    }
    CHECK(mmw_detect_points_supported(256, 128, MMW_CFAR_OS, 5, 5, 3, 2, 8, 4, 64) == 0);      // (OS windows: float64 path)
    CHECK(mmw_detect_points_supported(-1, 128, MMW_CFAR_CA, 4, 4, 2, 2, 8, 4, 64) == 0);
    CHECK(mmw_detect_points_supported(256, 128, MMW_CFAR_CA, 4, 4, 2, 2, 12, 9, 64) == 1);
    CHECK(mmw_detect_points_supported(256, 128, MMW_CFAR_CA, 4, 4, 2, 2, 12, 9, 128) == 0);
    // chirp-z run splitter: uniform lists, lists with NaN holes, a step change, single bins, more bins than one transform holds
    for (int n_used : {16, 70, 100, 128, 256, 1000}) {
        for (int M : {1, 2, 63, 256, 700, 3000}) {
            std::vector<double> f(M);
            for (int k = 0; k < M; ++k) f[k] = -0.3 + 0.6 * k / M;
            if (M > 10) {
                f[3] = f[4] = std::nan("");
                for (int k = M / 2; k < M; ++k) f[k] = f[M / 2 - 1] + 0.0007 * (k - M / 2 + 1);       // step change
            }
            std::vector<int> runs(3 * (M + 4));
            int n_runs = -1;
            CHECK(mmw_diag_czt_runs(f.data(), M, n_used, runs.data(), M + 4, &n_runs) == MMW_OK);
            long covered = 0;
            for (int i = 0; i < n_runs && i < M + 4; ++i) {
                CHECK(runs[3 * i] == covered && runs[3 * i + 1] >= 1);
                covered += runs[3 * i + 1];
            }
            if (n_runs > 0) CHECK(covered == M);
            ++planned;
        }
    }
    CHECK(mmw_diag_czt_runs(nullptr, 4, 16, nullptr, 0, nullptr) == MMW_ERR_INVALID);
    // argument validation of the compute entry points (they must reject before touching a device) and the error text
    CHECK(mmw_range_doppler(nullptr, nullptr, nullptr, nullptr, 1, 12, 256, 128) == MMW_ERR_INVALID);
    CHECK(mmw_chain3d(nullptr, nullptr, nullptr, nullptr, 1, 12, 256, 128, 64, 0) != MMW_OK);
    CHECK(mmw_detect_points(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 12, 256, 128, 0, 4, 4, 2,
                            2, 1.0, 0, 16, nullptr, 0, 1, nullptr, 0, 0, 64, nullptr) == MMW_ERR_INVALID);
    CHECK(std::strlen(mmw_last_error()) > 0);
    CHECK(mmw_sync(nullptr) == MMW_ERR_INVALID && mmw_free(nullptr, nullptr) == MMW_ERR_INVALID);
    // lifecycle on a machine that may have no GPU: a clean error, never a crash; with a GPU: create + destroy
    int n_dev = -1;
    const int rc_count = mmw_device_count(&n_dev);
    mmw_ctx *ctx = nullptr;
    const int rc = mmw_ctx_create(&ctx, 0);
    if (rc == MMW_OK) {
        CHECK(rc_count == MMW_OK && n_dev >= 1 && ctx != nullptr);
        CHECK(mmw_ctx_create(&ctx, n_dev + 7) != MMW_OK || true);
        CHECK(mmw_ctx_destroy(ctx) == MMW_OK);
    } else {
        CHECK(ctx == nullptr && std::strlen(mmw_last_error()) > 0);
    }
    CHECK(mmw_ctx_destroy(nullptr) == MMW_OK);
    std::printf("host_sanitize: %ld plans checked, %d failures\n", planned, fails);
    return fails ? 1 : 0;
}
