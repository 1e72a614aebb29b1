// Launch runtime of the learner step: process-owned HIP streams and HIP-graph segments.
//
// Role in the reference: TFAgent.train runs one cached concrete function per call
// (tfagent.py:56-70, the tf.function cache :60-64); here the step is captured once into
// HIP-graph segments and replayed.  The library owns the streams and the graph executables so
// that their lifetime is explicit:
//   * streams come from hipStreamCreateWithFlags, one per role - never a handle of a shared
//     round-robin pool, so a stream that captures is never a stream another thread issues on;
//   * a graph executable lives until dd_graph_destroy; the host side (graphs.py) never calls it
//     while the process runs (round-2 crash: executables dropped by one agent, then capture +
//     replay by the next agent died inside hipGraphLaunch).
#include "dd_common.h"
#include <vector>
#include "../../include/daydreamer_hip.h"
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

#define DD_HIP(call, where)                                  \
  do {                                                       \
    hipError_t e__ = (call);                                 \
    if (e__ != hipSuccess) {                                 \
      dd_set_error(where, e__);                              \
      return (int)e__;                                       \
    }                                                        \
  } while (0)

extern "C" int dd_stream_create(void** stream_out) {
  DD_REQUIRE(stream_out != nullptr, "dd_stream_create: null output");
  hipStream_t s = nullptr;
  DD_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "dd_stream_create");
  *stream_out = (void*)s;
  return 0;
}

extern "C" int dd_stream_destroy(void* stream) {
  DD_HIP(hipStreamDestroy((hipStream_t)stream), "dd_stream_destroy");
  return 0;
}

// Thread-local capture mode: other threads of the process (the minibatch prefetch thread, the
// RCCL watchdog) may call the runtime while this thread captures.
extern "C" int dd_graph_capture_begin(void* stream) {
  DD_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal),
         "dd_graph_capture_begin");
  return 0;
}

// Ends the capture on `stream` and instantiates it.  *exec_out = NULL (and *nodes_out = 0) for
// a segment without nodes (two adjacent cut points).  The graph template is released here; the
// executable keeps its own copy.
extern "C" int dd_graph_capture_end(void* stream, void** exec_out, int* nodes_out) {
  DD_REQUIRE(exec_out != nullptr, "dd_graph_capture_end: null output");
  *exec_out = nullptr;
  if (nodes_out) *nodes_out = 0;
  hipGraph_t graph = nullptr;
  DD_HIP(hipStreamEndCapture((hipStream_t)stream, &graph), "dd_graph_capture_end");
  DD_REQUIRE(graph != nullptr, "dd_graph_capture_end: no graph (capture invalidated)");
  size_t n = 0;
  hipError_t e = hipGraphGetNodes(graph, nullptr, &n);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(graph);
    dd_set_error("dd_graph_capture_end(nodes)", e);
    return (int)e;
  }
  if (nodes_out) *nodes_out = (int)n;
  // A captured segment holds kernel launches only (and the empty nodes of stream joins).  A memset
  // or copy node is refused: replayed next to a second process on the GPU, a memset node in front
  // of a kernel was not reliably complete before it (the fused observe scan's barrier counters,
  // docs/LABLOG.md end of round 6) - such work belongs in a kernel.  DD_GRAPH_ALLOW_NONKERNEL=1
  // turns the refusal into a message on stderr.
  if (n > 0) {
    std::vector<hipGraphNode_t> nodes(n);
    size_t m = n;
    e = hipGraphGetNodes(graph, nodes.data(), &m);
    int other = 0, first_type = -1;
    for (size_t i = 0; e == hipSuccess && i < m; ++i) {
      hipGraphNodeType ty = hipGraphNodeTypeKernel;
      e = hipGraphNodeGetType(nodes[i], &ty);
      if (e == hipSuccess && ty != hipGraphNodeTypeKernel && ty != hipGraphNodeTypeEmpty) {
        if (!other) first_type = (int)ty;
        ++other;
      }
    }
    if (e != hipSuccess) {
      (void)hipGraphDestroy(graph);
      dd_set_error("dd_graph_capture_end(node types)", e);
      return (int)e;
    }
    if (other) {
      static const int allow = getenv("DD_GRAPH_ALLOW_NONKERNEL") ? atoi(getenv("DD_GRAPH_ALLOW_NONKERNEL")) : 0;
      char msg[160];
      snprintf(msg, sizeof(msg), "dd_graph_capture_end: %d of %zu captured nodes are not kernel launches (first: hipGraphNodeType %d)",
               other, m, first_type);
      if (!allow) {
        (void)hipGraphDestroy(graph);
        DD_REQUIRE(false, msg);
      }
      fprintf(stderr, "libdaydreamer_hip: %s\n", msg);
    }
  }
  if (n > 0) {
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
      (void)hipGraphDestroy(graph);
      dd_set_error("dd_graph_capture_end(instantiate)", e);
      return (int)e;
    }
    *exec_out = (void*)exec;
  }
  DD_HIP(hipGraphDestroy(graph), "dd_graph_capture_end(destroy template)");
  return 0;
}

extern "C" int dd_graph_launch(void* exec, void* stream) {
  if (exec == nullptr) return 0;
  DD_HIP(hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)stream), "dd_graph_launch");
  return 0;
}

// Only safe once nothing that ran the executable is in flight (device synchronised).
extern "C" int dd_graph_destroy(void* exec) {
  if (exec == nullptr) return 0;
  DD_HIP(hipGraphExecDestroy((hipGraphExec_t)exec), "dd_graph_destroy");
  return 0;
}

// ---- native backtrace on a fatal signal -------------------------------------------------
// Python's faulthandler prints Python frames only; a crash inside the HIP runtime needs the
// native ones.  The handler writes them to stderr (async-signal-safe calls only) and hands
// over to the handler that was installed before (faulthandler, or the default action).

namespace {
struct sigaction g_prev[32];
const int g_sigs[] = {SIGSEGV, SIGBUS, SIGABRT, SIGFPE, SIGILL};
volatile sig_atomic_t g_in_handler = 0;

void crash_handler(int sig, siginfo_t* info, void* ctx) {
  if (!g_in_handler) {
    g_in_handler = 1;
    const char head[] = "\n==== libdaydreamer_hip: fatal signal, native backtrace ====\n";
    (void)!write(2, head, sizeof(head) - 1);
    void* frames[96];
    const int n = backtrace(frames, 96);
    backtrace_symbols_fd(frames, n, 2);
    const char tail[] = "==== end of native backtrace ====\n";
    (void)!write(2, tail, sizeof(tail) - 1);
  }
  const struct sigaction& prev = g_prev[sig];
  if ((prev.sa_flags & SA_SIGINFO) && prev.sa_sigaction) {
    prev.sa_sigaction(sig, info, ctx);
    return;
  }
  if (prev.sa_handler != SIG_DFL && prev.sa_handler != SIG_IGN && prev.sa_handler) {
    prev.sa_handler(sig);
    return;
  }
  signal(sig, SIG_DFL);
  raise(sig);
}
}  // namespace

extern "C" int dd_install_crash_handler(void) {
  static bool installed = false;
  if (installed) return 0;
  void* warm[4];
  backtrace(warm, 4);  // loads libgcc now: not async-signal-safe on first use
  for (int sig : g_sigs) {
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = crash_handler;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    if (sigaction(sig, &sa, &g_prev[sig]) != 0) {
      dd_set_error_msg("dd_install_crash_handler: sigaction failed");
      return -1;
    }
  }
  installed = true;
  return 0;
}
