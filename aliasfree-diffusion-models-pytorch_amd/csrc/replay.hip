// replay.hip -- a captured step re-issued on TWO REAL STREAMS from a C++ loop (host code only).
//
// Why: an eager train step costs ~5.6 ms of interpreter time against 6.7-7.0 ms of device time, so on a loaded host it becomes
// launch-bound; a hipGraph replay has no host cost but pays 0.2-0.3 ms for its cross-branch edges.  This
// executor takes the hipGraph that torch captured for the step (torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph()), walks its
// nodes ONCE -- kernel / memset / memcpy parameters, dependencies -- puts every node on one of two lanes (the weight-gradient
// kernels, recognised by name, on the side lane; everything else on the main lane) and turns the dependencies that cross lanes
// into event record / wait pairs.  A replay is then one pass over that list: hipLaunchKernel on the lane's stream, ~2 us per node,
// with exactly the stream semantics of the eager step.  The graph (and torch's private memory pool behind it) must stay alive.
#include <algorithm>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>
#include "common.h"

namespace afd {

struct ReplayNode {
  int type = -1;                       // hipGraphNodeType of a work node (kernel / memset / memcpy)
  int lane = 0;                        // 0 main, 1 side
  hipKernelNodeParams k{};
  hipMemsetParams ms{};
  hipMemcpy3DParms mc{};
  std::vector<int> wait;               // work nodes of the OTHER lane this node waits for (event of that node)
  hipEvent_t ev = nullptr;             // recorded after this node when a node of the other lane waits for it
  bool skip = false;                   // timing experiments only (AFD_REPLAY_SKIP): the node is not launched, its event still is
  bool dup = false;                    // timing experiments only (AFD_REPLAY_DUP): the node is launched twice (same inputs, same outputs)
};
struct ReplayPlan {
  std::vector<ReplayNode> nodes;       // work nodes in capture (= a topological) order
  int n_main = 0, n_side = 0, n_edges = 0;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;   // the side lane starts behind the main stream's current tail and the main stream ends behind the side lane's
};

static bool side_kernel(const char* name) {
  if (!name) return false;
  static const char* pat[] = {"wgrad", "fold_batched_k", "ln_c_bwd_plane", "silu_linear_dw_k", "colsum"};
  for (const char* p : pat)
    if (strstr(name, p)) return true;
  return false;
}

}  // namespace afd
using namespace afd;

extern "C" {

int afd_replay_build(void* hip_graph, void** out_handle, int* counts) {
  AFD_REQUIRE(hip_graph && out_handle, "afd_replay_build: null argument");
  hipGraph_t g = static_cast<hipGraph_t>(hip_graph);
  size_t n = 0;
  if (hipGraphGetNodes(g, nullptr, &n) != hipSuccess || n == 0) return set_error(AFD_EINVAL, "afd_replay_build: hipGraphGetNodes failed or empty graph");
  std::vector<hipGraphNode_t> gn(n);
  if (hipGraphGetNodes(g, gn.data(), &n) != hipSuccess) return set_error(AFD_EINVAL, "afd_replay_build: hipGraphGetNodes failed");
  std::unordered_map<hipGraphNode_t, int> index;
  for (size_t i = 0; i < n; ++i) index[gn[i]] = (int)i;
  // dependencies, node types
  std::vector<std::vector<int>> deps(n);
  std::vector<hipGraphNodeType> type(n);
  for (size_t i = 0; i < n; ++i) {
    if (hipGraphNodeGetType(gn[i], &type[i]) != hipSuccess) return set_error(AFD_EINVAL, "afd_replay_build: hipGraphNodeGetType failed");
    size_t nd = 0;
    if (hipGraphNodeGetDependencies(gn[i], nullptr, &nd) != hipSuccess) return set_error(AFD_EINVAL, "afd_replay_build: dependencies");
    std::vector<hipGraphNode_t> d(nd);
    if (nd && hipGraphNodeGetDependencies(gn[i], d.data(), &nd) != hipSuccess) return set_error(AFD_EINVAL, "afd_replay_build: dependencies");
    for (size_t j = 0; j < nd; ++j) {
      auto it = index.find(d[j]);
      if (it == index.end()) return set_error(AFD_EINVAL, "afd_replay_build: a dependency outside the graph");
      deps[i].push_back(it->second);
    }
  }
  // a topological order (the API's order is not promised to be one): Kahn, ties by the API's index (capture order)
  std::vector<int> order, indeg(n, 0);
  std::vector<std::vector<int>> succ(n);
  for (size_t i = 0; i < n; ++i)
    for (int d : deps[i]) { succ[d].push_back((int)i); ++indeg[i]; }
  {
    std::vector<int> ready;
    for (size_t i = 0; i < n; ++i)
      if (!indeg[i]) ready.push_back((int)i);
    while (!ready.empty()) {
      auto it = std::min_element(ready.begin(), ready.end());
      const int v = *it;
      ready.erase(it);
      order.push_back(v);
      for (int s : succ[v])
        if (--indeg[s] == 0) ready.push_back(s);
    }
    if (order.size() != n) return set_error(AFD_EINVAL, "afd_replay_build: the graph has a cycle");
  }
  auto plan = new ReplayPlan();
  std::vector<int> work_of(n, -1);                    // graph node -> index into plan->nodes (-1: no work)
  std::vector<std::vector<int>> eff(n);               // graph node -> the WORK nodes its completion implies (through no-op nodes)
  for (int v : order) {
    std::vector<int> e;
    for (int d : deps[v]) {
      if (work_of[d] >= 0) e.push_back(work_of[d]);
      else e.insert(e.end(), eff[d].begin(), eff[d].end());
    }
    std::sort(e.begin(), e.end());
    e.erase(std::unique(e.begin(), e.end()), e.end());
    const hipGraphNodeType t = type[v];
    if (t == hipGraphNodeTypeEmpty || t == hipGraphNodeTypeWaitEvent || t == hipGraphNodeTypeEventRecord) { eff[v] = e; continue; }
    ReplayNode r;
    r.type = (int)t;
    if (t == hipGraphNodeTypeKernel) {
      if (hipGraphKernelNodeGetParams(gn[v], &r.k) != hipSuccess) { delete plan; return set_error(AFD_EINVAL, "afd_replay_build: kernel node parameters"); }
      if (r.k.extra != nullptr && r.k.kernelParams == nullptr) { delete plan; return set_error(AFD_EINVAL, "afd_replay_build: a kernel node with packed (`extra`) arguments"); }
      const char* kname = hipKernelNameRefByPtr(r.k.func, nullptr);
      r.lane = side_kernel(kname) ? 1 : 0;
      // AFD_REPLAY_SKIP = comma-separated substrings of kernel names: those nodes are dropped from the replay (wrong results: what a
      // kernel family costs the step IN SITU is the step time with it against without it -- tools/replay_marginal.py)
      // AFD_REPLAY_DUP: the matching nodes are launched TWICE -- the data every other kernel sees stays what it was, so the step
      // time with the duplicate minus the step time without is the family's true cost in situ (dropping a family feeds stale or
      // zero data to everything behind it, and kernels run faster on zeros)
      if (const char* dp = getenv("AFD_REPLAY_DUP")) {
        std::string pats(dp);
        size_t b0 = 0;
        while (kname && b0 <= pats.size()) {
          const size_t e0 = pats.find(',', b0);
          const std::string pat = pats.substr(b0, e0 == std::string::npos ? std::string::npos : e0 - b0);
          if (!pat.empty() && strstr(kname, pat.c_str())) { r.dup = true; break; }
          if (e0 == std::string::npos) break;
          b0 = e0 + 1;
        }
      }
      if (const char* sk = getenv("AFD_REPLAY_SKIP")) {
        std::string pats(sk);
        size_t b0 = 0;
        while (kname && b0 <= pats.size()) {
          const size_t e0 = pats.find(',', b0);
          const std::string pat = pats.substr(b0, e0 == std::string::npos ? std::string::npos : e0 - b0);
          if (!pat.empty() && strstr(kname, pat.c_str())) { r.skip = true; break; }
          if (e0 == std::string::npos) break;
          b0 = e0 + 1;
        }
      }
    } else if (t == hipGraphNodeTypeMemset) {
      if (hipGraphMemsetNodeGetParams(gn[v], &r.ms) != hipSuccess) { delete plan; return set_error(AFD_EINVAL, "afd_replay_build: memset node parameters"); }
      if (r.ms.height > 1) { delete plan; return set_error(AFD_EINVAL, "afd_replay_build: a 2-D memset node"); }
    } else if (t == hipGraphNodeTypeMemcpy) {
      if (hipGraphMemcpyNodeGetParams(gn[v], &r.mc) != hipSuccess) { delete plan; return set_error(AFD_EINVAL, "afd_replay_build: memcpy node parameters"); }
      if (r.mc.extent.height > 1 || r.mc.extent.depth > 1 || r.mc.srcArray || r.mc.dstArray) { delete plan; return set_error(AFD_EINVAL, "afd_replay_build: a memcpy node that is not a flat copy"); }
    } else {
      delete plan;
      return set_error(AFD_EINVAL, "afd_replay_build: node type %d is not supported", (int)t);
    }
    const int me = (int)plan->nodes.size();
    for (int d : e)
      if (plan->nodes[d].lane != r.lane) r.wait.push_back(d);
    work_of[v] = me;
    plan->nodes.push_back(std::move(r));
  }
  // drop waits that an in-order lane already implies (an earlier node of MY lane waited for the same or a later node of the other
  // lane), create the events that are left
  int last_waited[2] = {-1, -1};                       // per lane: the latest other-lane node some node of this lane has waited for
  for (auto& r : plan->nodes) {
    std::vector<int> keep;
    std::sort(r.wait.begin(), r.wait.end());
    if (!r.wait.empty()) {
      const int d = r.wait.back();                     // the latest one implies the earlier ones (the other lane is in order)
      if (d > last_waited[r.lane]) { keep.push_back(d); last_waited[r.lane] = d; }
    }
    r.wait = keep;
    for (int d : r.wait) {
      if (!plan->nodes[d].ev && hipEventCreateWithFlags(&plan->nodes[d].ev, hipEventDisableTiming) != hipSuccess) {
        delete plan;
        return set_error(AFD_ELAUNCH, "afd_replay_build: hipEventCreate failed");
      }
      ++plan->n_edges;
    }
    (r.lane ? plan->n_side : plan->n_main)++;
  }
  if (hipEventCreateWithFlags(&plan->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&plan->ev_join, hipEventDisableTiming) != hipSuccess) {
    delete plan;
    return set_error(AFD_ELAUNCH, "afd_replay_build: hipEventCreate failed");
  }
  *out_handle = plan;
  if (counts) { counts[0] = (int)plan->nodes.size(); counts[1] = plan->n_main; counts[2] = plan->n_side; counts[3] = plan->n_edges; }
  return AFD_OK;
}

int afd_replay_run(void* handle, afd_stream_t main_stream, afd_stream_t side_stream) {
  AFD_REQUIRE(handle, "afd_replay_run: null handle");
  auto plan = static_cast<ReplayPlan*>(handle);
  hipStream_t st[2] = {as_stream(main_stream), as_stream(side_stream)};
  // the side lane starts behind whatever the main stream holds already (the inputs' copies) ...
  if (plan->n_side) {
    if (hipEventRecord(plan->ev_fork, st[0]) != hipSuccess || hipStreamWaitEvent(st[1], plan->ev_fork, 0) != hipSuccess)
      return set_error(AFD_ELAUNCH, "afd_replay_run: fork failed");
  }
  for (auto& r : plan->nodes) {
    hipStream_t s = st[r.lane];
    for (int d : r.wait)
      if (hipStreamWaitEvent(s, plan->nodes[d].ev, 0) != hipSuccess) return set_error(AFD_ELAUNCH, "afd_replay_run: hipStreamWaitEvent failed");
    hipError_t e = hipSuccess;
    if (r.skip) e = hipSuccess;
    else if (r.type == (int)hipGraphNodeTypeKernel) {
      e = hipLaunchKernel(r.k.func, r.k.gridDim, r.k.blockDim, r.k.kernelParams, r.k.sharedMemBytes, s);
      if (r.dup && e == hipSuccess) e = hipLaunchKernel(r.k.func, r.k.gridDim, r.k.blockDim, r.k.kernelParams, r.k.sharedMemBytes, s);
    }
    else if (r.type == (int)hipGraphNodeTypeMemset) {
      const size_t bytes = r.ms.width * r.ms.elementSize;
      if (r.ms.elementSize == 4) e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(r.ms.dst), (int)r.ms.value, r.ms.width, s);
      else if (r.ms.elementSize == 1) e = hipMemsetAsync(r.ms.dst, (int)r.ms.value, bytes, s);
      else if (r.ms.value == 0) e = hipMemsetAsync(r.ms.dst, 0, bytes, s);
      else return set_error(AFD_EINVAL, "afd_replay_run: memset with element size %u", r.ms.elementSize);
    } else {
      const char* src = static_cast<const char*>(r.mc.srcPtr.ptr) + r.mc.srcPos.x;
      char* dst = static_cast<char*>(r.mc.dstPtr.ptr) + r.mc.dstPos.x;
      e = hipMemcpyAsync(dst, src, r.mc.extent.width, hipMemcpyDefault, s);
    }
    if (e != hipSuccess) return set_error(AFD_ELAUNCH, "afd_replay_run: node launch failed: %s", hipGetErrorString(e));
    if (r.ev && hipEventRecord(r.ev, s) != hipSuccess) return set_error(AFD_ELAUNCH, "afd_replay_run: hipEventRecord failed");
  }
  // ... and the main stream ends behind the side lane's tail: a graph whose last nodes are the joined weight gradients (the
  // data-parallel form: the exchange and AdamW follow outside) has no node that carries that dependency
  if (plan->n_side) {
    if (hipEventRecord(plan->ev_join, st[1]) != hipSuccess || hipStreamWaitEvent(st[0], plan->ev_join, 0) != hipSuccess)
      return set_error(AFD_ELAUNCH, "afd_replay_run: join failed");
  }
  return AFD_OK;
}

int afd_replay_free(void* handle) {
  if (!handle) return AFD_OK;
  auto plan = static_cast<ReplayPlan*>(handle);
  for (auto& r : plan->nodes)
    if (r.ev) (void)hipEventDestroy(r.ev);
  if (plan->ev_fork) (void)hipEventDestroy(plan->ev_fork);
  if (plan->ev_join) (void)hipEventDestroy(plan->ev_join);
  delete plan;
  return AFD_OK;
}

}  // extern "C"
