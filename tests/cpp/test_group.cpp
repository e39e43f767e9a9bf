// GPU test program (run by tests/test_gpu_group.py): the multi-GPU C ABI (include/ismpc_group.h) used the way a C++ caller of the
// reference's kind would use it -- `new MPCSolver(ref)` once per GPU (Controller.cpp:105-106) becomes ismpc_group_create, the
// per-tick solve() (:346-348) becomes ismpc_group_solve_batch / ismpc_group_step_device -- on the ONE GPU of the test box:
// a group of one device (ncclCommInitAll), a group built from a unique id (ncclCommInitRank, world 1), the double-buffered device
// path, and the Formulation A group.  Every result must equal the plain handle's bytes.
//   usage: test_group <tick_in.bin> <batch> <a_state.bin> <a_push.bin> <a_batch>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ismpc_group.h"

#define CHECK(cond, ...) do { if (!(cond)) { std::fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); std::fprintf(stderr, __VA_ARGS__); \
    std::fprintf(stderr, " [group: %s] [ismpc: %s] [a: %s]\n", ismpc_group_last_error(), ismpc_last_error(), ismpc_a_last_error()); return 1; } } while (0)

template <typename T> static bool read_file(const char* path, std::vector<T>& v, size_t n)
{
    v.resize(n);
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    const size_t got = std::fread(v.data(), sizeof(T), n, f);
    std::fclose(f);
    return got == n;
}

int main(int argc, char** argv)
{
    if (argc < 6) { std::fprintf(stderr, "usage\n"); return 2; }
    const int batch = std::atoi(argv[2]), abatch = std::atoi(argv[5]);
    std::vector<ismpc_tick_in> in;
    CHECK(read_file(argv[1], in, (size_t)batch), "reading %s", argv[1]);

    // ---- the plan of Controller.cpp:89-97 and one plain handle as the yardstick
    ismpc_params p; ismpc_params_default(&p);
    const int rows = 40;
    std::vector<double> ftsp((size_t)rows * 4, 0.0);
    for (int i = 1; i < rows; ++i) {
        ftsp[4 * i + 0] = (i - 1) * 0.2; ftsp[4 * i + 1] = ((i - 1) % 2 == 0 ? 1.0 : -1.0) * 0.08;
        ftsp[4 * i + 3] = (p.mpc_dt / p.control_dt) * (p.S + p.F) * i;
    }
    ismpc_handle* h = nullptr;
    CHECK(ismpc_create(&p, ftsp.data(), rows, 0, &h) == ISMPC_OK, "ismpc_create");
    std::vector<ismpc_tick_out> ref((size_t)batch), out((size_t)batch);
    CHECK(ismpc_solve_batch(h, batch, in.data(), ref.data()) == ISMPC_OK, "ismpc_solve_batch");

    // ---- (i) one process, n = 1 device
    const int devices[1] = {0};
    ismpc_group* g = nullptr;
    CHECK(ismpc_group_create(&p, ftsp.data(), rows, devices, 1, &g) == ISMPC_OK, "ismpc_group_create");
    CHECK(ismpc_group_world(g) == 1 && ismpc_group_local(g) == 1 && ismpc_group_rank(g, 0) == 0, "world/local/rank");
    std::memset(out.data(), 0xab, sizeof(ismpc_tick_out) * (size_t)batch);
    CHECK(ismpc_group_solve_batch(g, batch, in.data(), out.data()) == ISMPC_OK, "ismpc_group_solve_batch");
    CHECK(std::memcmp(out.data(), ref.data(), sizeof(ismpc_tick_out) * (size_t)batch) == 0, "group of one differs from the plain handle");
    // a second, smaller and odd-sized batch through the same group (buffers reused)
    const int small = batch / 3 + 1;
    std::memset(out.data(), 0xab, sizeof(ismpc_tick_out) * (size_t)batch);
    CHECK(ismpc_group_solve_batch(g, small, in.data(), out.data()) == ISMPC_OK, "small batch");
    {   // (another batch-size class than the full batch: the kernels use another lane layout there, so the yardstick is the plain handle on the SAME batch)
        std::vector<ismpc_tick_out> ref_small((size_t)small);
        CHECK(ismpc_solve_batch(h, small, in.data(), ref_small.data()) == ISMPC_OK, "ismpc_solve_batch (small)");
        CHECK(std::memcmp(out.data(), ref_small.data(), sizeof(ismpc_tick_out) * (size_t)small) == 0, "small batch differs");
    }

    // ---- the device path, double-buffered: 6 steps, the input of step k is the batch rotated by k records
    ismpc_tick_in* d_in = nullptr;
    CHECK(hipMalloc((void**)&d_in, sizeof(ismpc_tick_in) * (size_t)batch * 2) == hipSuccess, "hipMalloc");
    CHECK(hipMemcpy(d_in, in.data(), sizeof(ismpc_tick_in) * (size_t)batch, hipMemcpyHostToDevice) == hipSuccess, "H2D");
    CHECK(hipMemcpy(d_in + batch, in.data(), sizeof(ismpc_tick_in) * (size_t)batch, hipMemcpyHostToDevice) == hipSuccess, "H2D");
    CHECK(ismpc_group_reserve(g, batch) == ISMPC_OK, "reserve");
    std::vector<ismpc_tick_out> got((size_t)batch);
    for (int k = 0; k < 6; ++k) {
        const ismpc_tick_in* shard[1] = {d_in + k};                       // world 1: the shard is the whole batch
        CHECK(ismpc_group_step_device(g, batch, shard, k & 1) == ISMPC_OK, "step %d", k);
        if (k >= 1) {                                                     // read step k-1's gathered records while step k runs
            ismpc_tick_out* res = nullptr;
            CHECK(ismpc_group_result_device(g, 0, (k - 1) & 1, &res) == ISMPC_OK && res, "result");
            hipStream_t s; CHECK(hipStreamCreate(&s) == hipSuccess, "stream");
            CHECK(ismpc_group_wait_on(g, 0, (k - 1) & 1, s) == ISMPC_OK, "wait_on");
            CHECK(hipMemcpyAsync(got.data(), res, sizeof(ismpc_tick_out) * (size_t)batch, hipMemcpyDeviceToHost, s) == hipSuccess, "D2H");
            CHECK(hipStreamSynchronize(s) == hipSuccess, "sync"); (void)hipStreamDestroy(s);
            // step k-1 solved records (k-1) .. (k-1)+batch-1 of the doubled input = ref rotated by k-1
            for (int i = 0; i < batch; ++i)
                CHECK(std::memcmp(&got[i], &ref[(i + k - 1) % batch], sizeof(ismpc_tick_out)) == 0, "step %d record %d", k - 1, i);
        }
    }
    CHECK(ismpc_group_sync(g) == ISMPC_OK, "sync");
    ismpc_group_destroy(g);

    // ---- (ii) one process per GPU: a communicator from a unique id, world = 1
    unsigned char uid[ISMPC_UNIQUE_ID_BYTES];
    CHECK(ismpc_group_unique_id(uid) == ISMPC_OK, "unique id");
    ismpc_group* gr = nullptr;
    CHECK(ismpc_group_create_rank(&p, ftsp.data(), rows, 0, uid, 0, 1, &gr) == ISMPC_OK, "ismpc_group_create_rank");
    CHECK(ismpc_group_world(gr) == 1, "world");
    std::memset(out.data(), 0xab, sizeof(ismpc_tick_out) * (size_t)batch);
    CHECK(ismpc_group_solve_batch(gr, batch, in.data(), out.data()) == ISMPC_OK, "rank-mode solve");
    CHECK(std::memcmp(out.data(), ref.data(), sizeof(ismpc_tick_out) * (size_t)batch) == 0, "rank-mode group differs from the plain handle");
    ismpc_group_destroy(gr);
    // bad arguments are refused, not crashed on
    const int twice[2] = {0, 0};
    CHECK(ismpc_group_create(&p, ftsp.data(), rows, twice, 2, &gr) == ISMPC_E_INVALID && gr == nullptr, "a device twice must be refused");
    const int beyond[1] = {64};
    CHECK(ismpc_group_create(&p, ftsp.data(), rows, beyond, 1, &gr) == ISMPC_E_INVALID, "device ordinal beyond the box");

    // ---- Formulation A: one tick of a pushed walking-gait batch, group against plain handle
    std::vector<ismpc_a_state> st0; std::vector<double> push;
    CHECK(read_file(argv[3], st0, (size_t)abatch) && read_file(argv[4], push, (size_t)abatch * 2), "reading the A inputs");
    ismpc_a_gait gait; ismpc_a_gait_default(1, 0.78539816339744828, 0.1, &gait);
    ismpc_a_params ap; ismpc_a_params_default(1, &ap);
    std::vector<double> fp((size_t)(gait.n_gait + 1) * 8), ce((size_t)gait.n_gait * 2);
    CHECK(ismpc_a_plan(&gait, fp.data(), ce.data()) > 0, "ismpc_a_plan");
    ismpc_a_handle* ha = nullptr;
    CHECK(ismpc_a_create(&ap, ce.data(), 0, &ha) == ISMPC_OK, "ismpc_a_create");
    ismpc_a_state* d_st = nullptr; double* d_push = nullptr; ismpc_a_out* d_out = nullptr;
    CHECK(hipMalloc((void**)&d_st, sizeof(ismpc_a_state) * (size_t)abatch) == hipSuccess && hipMalloc((void**)&d_push, 16 * (size_t)abatch) == hipSuccess &&
          hipMalloc((void**)&d_out, sizeof(ismpc_a_out) * (size_t)abatch) == hipSuccess, "hipMalloc");
    CHECK(hipMemcpy(d_st, st0.data(), sizeof(ismpc_a_state) * (size_t)abatch, hipMemcpyHostToDevice) == hipSuccess, "H2D");
    CHECK(hipMemcpy(d_push, push.data(), 16 * (size_t)abatch, hipMemcpyHostToDevice) == hipSuccess, "H2D");
    CHECK(ismpc_a_tick_batch_device(ha, abatch, d_st, d_push, d_out, nullptr) == ISMPC_OK, "ismpc_a_tick_batch_device");
    std::vector<ismpc_a_out> aref((size_t)abatch), aout((size_t)abatch); std::vector<ismpc_a_state> sref((size_t)abatch), sgot = st0;
    CHECK(hipMemcpy(aref.data(), d_out, sizeof(ismpc_a_out) * (size_t)abatch, hipMemcpyDeviceToHost) == hipSuccess, "D2H");
    CHECK(hipMemcpy(sref.data(), d_st, sizeof(ismpc_a_state) * (size_t)abatch, hipMemcpyDeviceToHost) == hipSuccess, "D2H");
    ismpc_a_group* ga = nullptr;
    CHECK(ismpc_a_group_create(&ap, ce.data(), devices, 1, &ga) == ISMPC_OK, "ismpc_a_group_create");
    CHECK(ismpc_a_group_world(ga) == 1, "world");
    CHECK(ismpc_a_group_tick_batch(ga, abatch, sgot.data(), nullptr, push.data(), aout.data()) == ISMPC_OK, "ismpc_a_group_tick_batch");
    CHECK(std::memcmp(aout.data(), aref.data(), sizeof(ismpc_a_out) * (size_t)abatch) == 0, "A group records differ from the plain handle");
    CHECK(std::memcmp(sgot.data(), sref.data(), sizeof(ismpc_a_state) * (size_t)abatch) == 0, "A group state differs from the plain handle");
    ismpc_a_group_destroy(ga);
    ismpc_a_destroy(ha); ismpc_destroy(h);
    (void)hipFree(d_in); (void)hipFree(d_st); (void)hipFree(d_push); (void)hipFree(d_out);
    std::printf("OK world=1 rccl=%d ragged=%s\n", ismpc_group_rccl_version(), std::getenv("ISMPC_GROUP_FORCE_RAGGED") ? "forced" : "no");
    return 0;
}
