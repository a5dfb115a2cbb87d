// flvis_amd: context object behind the C ABI (include/flvis_hip.h).
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <memory>
#include <string>

#include "img_kernels.hpp"

namespace flvis {
struct Pipeline;  // full front-end + local-map state (pipeline.hpp); null for kernel-level-only contexts
}

struct flvis_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;

  struct Buf {
    void* p = nullptr;
    size_t bytes = 0;
  };
  std::map<std::string, Buf> bufs;  // named scratch buffers, grown on demand (never inside a steady-state loop)
  flvis::Pipeline* pipe = nullptr;
  int voc_nodes = 0, voc_words = 0, voc_depth = 0;  // the DBoW3 vocabulary resident in the `voc_*` scratch buffers (flvis_hip_bow_set_vocabulary)
  std::string orb_pattern;  // the BRIEF pattern currently resident in the `orb_pattern` scratch buffer (1024 bytes) or empty

  // returns a device buffer of at least `bytes` bytes (contents undefined after growth)
  void* scratch(const std::string& name, size_t bytes, bool zero_on_alloc = false);
  int fail(int code, const std::string& msg) {
    err = msg;
    return code;
  }
  int hip_fail(hipError_t e, const char* what) {
    err = std::string(what) + ": " + hipGetErrorString(e);
    return -3;
  }
};

static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }
