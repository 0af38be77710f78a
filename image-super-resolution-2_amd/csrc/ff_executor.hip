// C-level executor: replays a launch plan recorded from the Python host (isr2_amd/plan.py) -- every kernel of
// CompleteEnhancedFusionSR.forward in eval mode (src/models/enhanced_fusion.py:694-754, the call the reference plugin makes at
// models/team29_FreqFusion/io.py:221) for one input shape -- without Python in the loop:
//     ff_create(plan) -> ff_upload(slot, bytes) for every prepared-weight slot -> ff_finalize -> ff_forward(lr, out, stream) ...
// The library owns its weight copies and one workspace (packed by buffer lifetime at export time); the caller owns the input
// and output tensors; every launch goes to the caller's stream in plan order, so ff_forward is asynchronous and
// graph-capturable like the per-operator entry points it drives.  No internal threads; one handle per host thread.
#include "ff_common.h"
#include <stdint.h>
#include <stdlib.h>
#include <string>
#include <vector>

extern "C" int ff_abi_version(void);
union FFVal { long long i; double f; void* p; };
#include "ff_dispatch_gen.h"      // generated from include/ff_kernels.h by build.py: FF_DISPATCH_NAMES[], ff_dispatch(id, vals)

enum { K_INT = 0, K_FLT, K_NULL, K_WEIGHT, K_WORK, K_INPUT, K_OUTPUT, K_STREAM };

struct PlanArg { uint8_t kind; long long a, b; double f; };
struct PlanCall { int fn; int first, n; int stream; };      // fn == -1: fork (stream 0) / join (stream 1) marker
struct PlanSlot { std::string name; long long nbytes; void* dev; bool filled; };
struct ff_model {
  std::vector<PlanSlot> slots;
  std::vector<PlanCall> calls;
  std::vector<PlanArg> args;
  long long wbytes = 0;
  void* work = nullptr;
  int in[4], out[4];
  bool finalized = false;
  // the host's three-stream schedule: between a fork and a join marker the launches of stream 1 / 2 go to two internal streams
  hipStream_t side[2] = {nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
  int nstreams = 1;
};

static bool rd(FILE* f, void* dst, size_t n) { return fread(dst, 1, n, f) == n; }

static void free_model(ff_model* m) {
  if (!m) return;
  for (auto& s : m->slots) if (s.dev) (void)hipFree(s.dev);
  if (m->work) (void)hipFree(m->work);
  for (int i = 0; i < 2; ++i) { if (m->side[i]) (void)hipStreamDestroy(m->side[i]); if (m->ev_join[i]) (void)hipEventDestroy(m->ev_join[i]); }
  if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
  delete m;
}

extern "C" int ff_create(const char* plan_path, void** out_handle) {
  FF_CHECK_ARG(plan_path && out_handle, "ff_create: null argument");
  *out_handle = nullptr;
  FILE* f = fopen(plan_path, "rb");
  FF_CHECK_ARG(f, "ff_create: cannot open %s", plan_path);
  ff_model* m = new ff_model();
  auto fail = [&](const char* why) { ff_set_error("ff_create: %s: %s", plan_path, why); fclose(f); free_model(m); return FF_ERR_ARG; };
  char magic[8];
  int32_t abi; uint32_t nnames;
  if (!rd(f, magic, 8) || memcmp(magic, "FFPLAN3\0", 8) != 0) return fail("not an FFPLAN3 file");
  if (!rd(f, &abi, 4) || !rd(f, &nnames, 4) || nnames > 4096) return fail("truncated header");
  if (abi != ff_abi_version()) return fail("plan was recorded against another ABI version of the kernel library");
  std::vector<int> fnmap(nnames, -1);
  for (uint32_t i = 0; i < nnames; ++i) {
    uint32_t ln;
    if (!rd(f, &ln, 4) || ln > 256) return fail("bad entry-point name");
    std::string nm(ln, '\0');
    if (!rd(f, &nm[0], ln)) return fail("truncated names");
    for (int k = 0; k < FF_DISPATCH_COUNT; ++k)
      if (nm == FF_DISPATCH_NAMES[k]) fnmap[i] = k;
    if (fnmap[i] < 0) { ff_set_error("ff_create: %s uses entry point %s, which this library does not export", plan_path, nm.c_str()); fclose(f); free_model(m); return FF_ERR_ARG; }
  }
  uint32_t nslots;
  if (!rd(f, &nslots, 4) || nslots > (1u << 20)) return fail("bad slot count");
  m->slots.resize(nslots);
  for (auto& s : m->slots) {
    uint32_t ln;
    if (!rd(f, &ln, 4) || ln > 1024) return fail("bad slot name");
    s.name.assign(ln, '\0');
    if (!rd(f, &s.name[0], ln) || !rd(f, &s.nbytes, 8) || s.nbytes < 0) return fail("truncated slot table");
    s.dev = nullptr; s.filled = false;
  }
  if (!rd(f, &m->wbytes, 8) || !rd(f, m->in, 16) || !rd(f, m->out, 16) || m->wbytes < 0) return fail("truncated shapes");
  uint32_t ncalls;
  if (!rd(f, &ncalls, 4) || ncalls > (1u << 24)) return fail("bad call count");
  m->calls.resize(ncalls);
  for (auto& c : m->calls) {
    uint16_t fid, na, sid;
    if (!rd(f, &fid, 2) || !rd(f, &na, 2) || !rd(f, &sid, 2)) return fail("bad call record");
    if (fid == 0xFFFF) { c.fn = -1; c.first = 0; c.n = 0; c.stream = sid; if (sid > 1) return fail("bad marker"); continue; }
    if (fid >= nnames || na > 64 || sid > 2) return fail("bad call record");
    c.fn = fnmap[fid]; c.first = (int)m->args.size(); c.n = na; c.stream = sid;
    const char* kinds = FF_DISPATCH_KINDS[c.fn];
    if ((size_t)na != strlen(kinds)) { ff_set_error("ff_create: %s: a call of %s carries %d arguments, the entry point takes %zu", plan_path, FF_DISPATCH_NAMES[c.fn], (int)na, strlen(kinds)); fclose(f); free_model(m); return FF_ERR_ARG; }
    if (sid + 1 > m->nstreams) m->nstreams = sid + 1;
    for (int i = 0; i < na; ++i) {
      unsigned char rec[20];
      if (!rd(f, rec, 20)) return fail("truncated arguments");
      PlanArg a; a.kind = rec[0]; a.a = 0; a.b = 0; a.f = 0.0;
      if (a.kind == K_FLT) memcpy(&a.f, rec + 4, 8);
      else { memcpy(&a.a, rec + 4, 8); memcpy(&a.b, rec + 12, 8); }
      if (a.kind > K_STREAM) return fail("bad argument kind");
      if (a.kind == K_WEIGHT && (a.a < 0 || a.a >= (long long)nslots || a.b < 0 || a.b >= m->slots[a.a].nbytes)) return fail("weight reference out of range");
      if (a.kind == K_WORK && (a.b < 0 || a.b >= m->wbytes)) return fail("workspace reference out of range");
      {
        const long long in_bytes = 4LL * m->in[0] * m->in[1] * m->in[2] * m->in[3], out_bytes = 4LL * m->out[0] * m->out[1] * m->out[2] * m->out[3];
        if (a.kind == K_INPUT && (a.b < 0 || a.b >= in_bytes)) return fail("input reference beyond the lr tensor");
        if (a.kind == K_OUTPUT && (a.b < 0 || a.b >= out_bytes)) return fail("output reference beyond the sr tensor");
        const char want = kinds[i];
        const bool ok = want == 'P' ? (a.kind == K_NULL || a.kind == K_WEIGHT || a.kind == K_WORK || a.kind == K_INPUT || a.kind == K_OUTPUT)
                       : want == 'I' ? a.kind == K_INT : want == 'F' ? a.kind == K_FLT : a.kind == K_STREAM;
        if (!ok) { ff_set_error("ff_create: %s: argument %d of %s has kind %d where the prototype has '%c'", plan_path, i, FF_DISPATCH_NAMES[c.fn], (int)a.kind, want); fclose(f); free_model(m); return FF_ERR_ARG; }
      }
      m->args.push_back(a);
    }
  }
  fclose(f);
  for (auto& s : m->slots)
    if (hipMalloc(&s.dev, (size_t)(s.nbytes > 0 ? s.nbytes : 1)) != hipSuccess) { ff_set_error("ff_create: hipMalloc of slot %s (%lld bytes) failed", s.name.c_str(), s.nbytes); free_model(m); return FF_ERR_LAUNCH; }
  if (hipMalloc(&m->work, (size_t)(m->wbytes > 0 ? m->wbytes : 1)) != hipSuccess) { ff_set_error("ff_create: hipMalloc of the %lld-byte workspace failed", m->wbytes); free_model(m); return FF_ERR_LAUNCH; }
  if (m->nstreams > 1) {
    bool ok = hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 2 && ok; ++i)
      ok = hipStreamCreateWithFlags(&m->side[i], hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&m->ev_join[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) { ff_set_error("ff_create: cannot create the executor's streams / events"); free_model(m); return FF_ERR_LAUNCH; }
  }
  *out_handle = m;
  return FF_OK;
}

extern "C" int ff_upload(void* handle, const char* slot_name, const void* host_or_dev_ptr, long long nbytes) {
  ff_model* m = (ff_model*)handle;
  FF_CHECK_ARG(m && slot_name && host_or_dev_ptr, "ff_upload: null argument");
  for (auto& s : m->slots)
    if (s.name == slot_name) {
      FF_CHECK_ARG(nbytes == s.nbytes, "ff_upload: slot %s holds %lld bytes, got %lld", slot_name, s.nbytes, nbytes);
      if (hipMemcpy(s.dev, host_or_dev_ptr, (size_t)nbytes, hipMemcpyDefault) != hipSuccess) { ff_set_error("ff_upload: copy into %s failed", slot_name); return FF_ERR_LAUNCH; }
      s.filled = true;
      return FF_OK;
    }
  ff_set_error("ff_upload: the plan has no slot named %s", slot_name);
  return FF_ERR_ARG;
}

extern "C" int ff_finalize(void* handle) {
  ff_model* m = (ff_model*)handle;
  FF_CHECK_ARG(m, "ff_finalize: null handle");
  for (auto& s : m->slots) FF_CHECK_ARG(s.filled, "ff_finalize: slot %s was never uploaded", s.name.c_str());
  m->finalized = true;
  return FF_OK;
}

extern "C" int ff_forward(void* handle, const float* lr_dev, int B, int H, int W, float* out_dev, void* stream) {
  ff_model* m = (ff_model*)handle;
  FF_CHECK_ARG(m && lr_dev && out_dev, "ff_forward: null argument");
  FF_CHECK_ARG(m->finalized, "ff_forward: ff_finalize has not succeeded on this handle");
  FF_CHECK_ARG(B == m->in[0] && H == m->in[2] && W == m->in[3], "ff_forward: this plan is for input [%d,3,%d,%d], got [%d,3,%d,%d]", m->in[0], m->in[2], m->in[3], B, H, W);
  FFVal v[64];
  hipStream_t main_st = (hipStream_t)stream;
  for (const PlanCall& c : m->calls) {
    if (c.fn < 0) {
      if (m->nstreams > 1) {
        bool ok = true;
        if (c.stream == 0) {                                 // fork: the side streams wait for everything issued on the caller's stream so far
          ok = hipEventRecord(m->ev_fork, main_st) == hipSuccess;
          for (int i = 0; i < 2 && ok; ++i) ok = hipStreamWaitEvent(m->side[i], m->ev_fork, 0) == hipSuccess;
        } else {                                             // join: the caller's stream waits for both side streams
          for (int i = 0; i < 2 && ok; ++i)
            ok = hipEventRecord(m->ev_join[i], m->side[i]) == hipSuccess && hipStreamWaitEvent(main_st, m->ev_join[i], 0) == hipSuccess;
        }
        if (!ok) { ff_set_error("ff_forward: stream fork / join failed: %s", hipGetErrorString(hipGetLastError())); return FF_ERR_LAUNCH; }
      }
      continue;
    }
    void* call_stream = (c.stream == 0 || m->nstreams == 1) ? stream : (void*)m->side[c.stream - 1];
    for (int i = 0; i < c.n; ++i) {
      const PlanArg& a = m->args[c.first + i];
      switch (a.kind) {
        case K_INT: v[i].i = a.a; break;
        case K_FLT: v[i].f = a.f; break;
        case K_NULL: v[i].p = nullptr; break;
        case K_WEIGHT: v[i].p = (char*)m->slots[a.a].dev + a.b; break;
        case K_WORK: v[i].p = (char*)m->work + a.b; break;
        case K_INPUT: v[i].p = (char*)lr_dev + a.b; break;
        case K_OUTPUT: v[i].p = (char*)out_dev + a.b; break;
        default: v[i].p = call_stream; break;
      }
    }
    const int rc = ff_dispatch(c.fn, v);
    if (rc != FF_OK) return rc;               // ff_last_error() carries the failing entry point's message
  }
  return FF_OK;
}

extern "C" int ff_destroy(void* handle) {
  free_model((ff_model*)handle);
  return FF_OK;
}

extern "C" int ff_model_io_shape(void* handle, int* in_shape4, int* out_shape4) {
  ff_model* m = (ff_model*)handle;
  FF_CHECK_ARG(m && in_shape4 && out_shape4, "ff_model_io_shape: null argument");
  for (int i = 0; i < 4; ++i) { in_shape4[i] = m->in[i]; out_shape4[i] = m->out[i]; }
  return FF_OK;
}
extern "C" int ff_model_num_slots(void* handle) { return handle ? (int)((ff_model*)handle)->slots.size() : -1; }
extern "C" const char* ff_model_slot_name(void* handle, int i) {
  ff_model* m = (ff_model*)handle;
  return (m && i >= 0 && i < (int)m->slots.size()) ? m->slots[i].name.c_str() : nullptr;
}
extern "C" long long ff_model_slot_bytes(void* handle, int i) {
  ff_model* m = (ff_model*)handle;
  return (m && i >= 0 && i < (int)m->slots.size()) ? m->slots[i].nbytes : -1;
}
extern "C" long long ff_model_workspace_bytes(void* handle) { return handle ? ((ff_model*)handle)->wbytes : -1; }
extern "C" int ff_model_num_launches(void* handle) { return handle ? (int)((ff_model*)handle)->calls.size() : -1; }
