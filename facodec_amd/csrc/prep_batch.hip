// Recorder / replayer of the batched weight preparation (prep_batch.h).  Host code only: the batch kernels live next to the single
// launches they mirror (pack.hip, conv1d_bsplit.hip, conv1d_gemm_split.hip, conv1d_bsplit2.hip, conv1d_bwd.hip).
#include "prep_batch.h"
#include <algorithm>
#include <mutex>
#include <vector>

namespace fac {

namespace {

struct Recorded { int phase, unit; PrepJob job; };

struct Group {              // one launch of a replay: the jobs of one (phase, unit)
  int phase, unit, njobs, total;
  const PrepJob* jobs;      // device
  const int* first;         // device
};

struct Plan {
  std::vector<Group> groups;
  void* dev = nullptr;      // one allocation: every group's job table and first-block table
  int njobs = 0;
  bool live = false;
};

thread_local bool t_recording = false;
thread_local int t_phase = 0;
thread_local std::vector<Recorded>* t_jobs = nullptr;

std::mutex g_mu;
std::vector<Plan> g_plans;

typedef int (*launch_fn)(const PrepJob*, const int*, int, int, hipStream_t);
const launch_fn g_launch[PU_COUNT] = {prep_launch_pack, prep_launch_bsplit, prep_launch_gsplit, prep_launch_bsplit2, prep_launch_bwd};

}  // namespace

bool prep_recording() { return t_recording; }

int prep_record(int unit, const PrepJob& j) {
  if (!t_recording || !t_jobs) {
    set_error("prep_record: not recording");
    return FAC_ERR_ARG;
  }
  if (j.nblocks <= 0) {
    set_error("prep_record: empty job");
    return FAC_ERR_ARG;
  }
  t_jobs->push_back(Recorded{t_phase, unit, j});
  return FAC_OK;
}

}  // namespace fac

extern "C" int fac_prep_begin(void) {
  using namespace fac;
  FAC_REQUIRE(!t_recording, "prep_begin: this thread is already recording");
  if (!t_jobs) t_jobs = new std::vector<Recorded>();
  t_jobs->clear();
  t_phase = 0;
  t_recording = true;
  return FAC_OK;
}

extern "C" int fac_prep_set_phase(int phase) {
  using namespace fac;
  FAC_REQUIRE(t_recording && phase >= 0, "prep_set_phase: not recording, or a negative phase");
  t_phase = phase;
  return FAC_OK;
}

extern "C" int fac_prep_abort(void) {
  using namespace fac;
  t_recording = false;
  if (t_jobs) t_jobs->clear();
  return FAC_OK;
}

// Ends the recording and builds the plan (device tables: one hipMalloc, one blocking copy -- not inside a stream capture).
// Returns the plan id (>= 0), or a negative error code.
extern "C" int fac_prep_end(void) {
  using namespace fac;
  if (!t_recording) {
    set_error("prep_end: not recording");
    return FAC_ERR_ARG;
  }
  t_recording = false;
  std::vector<Recorded> jobs;
  jobs.swap(*t_jobs);
  if (jobs.empty()) {
    set_error("prep_end: nothing was recorded");
    return FAC_ERR_ARG;
  }
  std::stable_sort(jobs.begin(), jobs.end(), [](const Recorded& x, const Recorded& y) {
    return x.phase != y.phase ? x.phase < y.phase : x.unit < y.unit;
  });
  // layout of the device buffer: [jobs of group 0][jobs of group 1]...[first tables], 16-byte aligned pieces
  const size_t nj = jobs.size();
  const size_t jobs_bytes = nj * sizeof(PrepJob);
  const size_t first_off = (jobs_bytes + 15) & ~(size_t)15;
  std::vector<unsigned char> host(first_off + nj * sizeof(int));
  PrepJob* hj = reinterpret_cast<PrepJob*>(host.data());
  int* hf = reinterpret_cast<int*>(host.data() + first_off);
  Plan plan;
  size_t i = 0;
  while (i < nj) {
    size_t e = i;
    long long total = 0;
    while (e < nj && jobs[e].phase == jobs[i].phase && jobs[e].unit == jobs[i].unit) {
      hj[e] = jobs[e].job;
      hf[e] = (int)total;
      total += jobs[e].job.nblocks;
      ++e;
    }
    if (total > 0x7fffffffll) {
      set_error("prep_end: more than 2^31 workgroups in one group");
      return FAC_ERR_ARG;
    }
    Group g{jobs[i].phase, jobs[i].unit, (int)(e - i), (int)total, nullptr, nullptr};
    g.jobs = reinterpret_cast<const PrepJob*>(i * sizeof(PrepJob));          // offsets, rebased below
    g.first = reinterpret_cast<const int*>(first_off + i * sizeof(int));
    plan.groups.push_back(g);
    i = e;
  }
  void* dev = nullptr;
  if (hipMalloc(&dev, host.size()) != hipSuccess || hipMemcpy(dev, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess) {
    if (dev) (void)hipFree(dev);
    set_error("prep_end: device table allocation / upload failed: %s", hipGetErrorString(hipGetLastError()));
    return FAC_ERR_LAUNCH;
  }
  for (Group& g : plan.groups) {
    g.jobs = reinterpret_cast<const PrepJob*>(static_cast<unsigned char*>(dev) + reinterpret_cast<size_t>(g.jobs));
    g.first = reinterpret_cast<const int*>(static_cast<unsigned char*>(dev) + reinterpret_cast<size_t>(g.first));
  }
  plan.dev = dev;
  plan.njobs = (int)nj;
  plan.live = true;
  std::lock_guard<std::mutex> lk(g_mu);
  for (size_t k = 0; k < g_plans.size(); ++k)
    if (!g_plans[k].live) {
      g_plans[k] = std::move(plan);
      return (int)k;
    }
  g_plans.push_back(std::move(plan));
  return (int)g_plans.size() - 1;
}

// Runs every recorded job again on `stream`, phases in order, one launch per (phase, translation unit).
extern "C" int fac_prep_replay(int plan, fac_stream_t stream) {
  using namespace fac;
  std::vector<Group> groups;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    FAC_REQUIRE(plan >= 0 && plan < (int)g_plans.size() && g_plans[plan].live, "prep_replay: no such plan");
    groups = g_plans[plan].groups;
  }
  for (const Group& g : groups) {
    const int rc = g_launch[g.unit](g.jobs, g.first, g.njobs, g.total, (hipStream_t)stream);
    if (rc != FAC_OK) return rc;
  }
  return FAC_OK;
}

// jobs / launches of a plan (tests, logs)
extern "C" int fac_prep_info(int plan, int* n_jobs, int* n_launches) {
  using namespace fac;
  std::lock_guard<std::mutex> lk(g_mu);
  FAC_REQUIRE(plan >= 0 && plan < (int)g_plans.size() && g_plans[plan].live, "prep_info: no such plan");
  if (n_jobs) *n_jobs = g_plans[plan].njobs;
  if (n_launches) *n_launches = (int)g_plans[plan].groups.size();
  return FAC_OK;
}

// The caller guarantees that no replay of the plan is still executing (its tables are freed).
extern "C" int fac_prep_free(int plan) {
  using namespace fac;
  std::lock_guard<std::mutex> lk(g_mu);
  FAC_REQUIRE(plan >= 0 && plan < (int)g_plans.size() && g_plans[plan].live, "prep_free: no such plan");
  (void)hipFree(g_plans[plan].dev);
  g_plans[plan] = Plan();
  return FAC_OK;
}
