// Handle, device query, hipGraph capture helpers and HIP-event timing.
#include "common.hpp"

extern "C" int tfmq_abi_version(void) { return 9; }   // 9: tfmq_adaround_scalars / tfmq_adaround_bwd_adam_dyn (captured reconstruction iterations, round 5); 8: tfmq_ff_fused, TFMQ_OUT_GEGLU_Q8_FAST (round 4); 7: tfmq_conv_desc.ksplit (split-K of the w4a8 tile kernel); 6: + Fisher-weighted reconstruction (upsample2x_bwd, kl_softmax_grad, fisher_loss); 3: tfmq_conv_desc grew x2 / cin1; 4: + w64 (round 2); 5: + W8A8 / attention-quantizer / GEMM-precision entry points, TFMQ_TILE_SLAB128 (round 3)

extern "C" int tfmq_create(int device, tfmq_handle* out) {
  if (!out) return TFMQ_ERR_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return TFMQ_ERR_HIP;
  if (hipSetDevice(device) != hipSuccess) return TFMQ_ERR_HIP;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, device) != hipSuccess) return TFMQ_ERR_HIP;
  tfmq_ctx* c = new tfmq_ctx();
  c->device = device;
  c->cu_count = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
  c->clock_khz = p.clockRate;
  c->hbm_bytes = p.totalGlobalMem;
  {
    std::vector<unsigned char> host(256 * 64);
    for (int v = 0; v < 256; ++v)
      for (int i = 0; i < 64; ++i) host[v * 64 + i] = static_cast<unsigned char>(v);
    if (hipMalloc(reinterpret_cast<void**>(&c->pad_table), host.size()) != hipSuccess ||
        hipMemcpy(c->pad_table, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess) {
      delete c;
      return TFMQ_ERR_HIP;
    }
  }
  if (hipMalloc(reinterpret_cast<void**>(&c->ksplit_ws), tfmq_ctx::KSPLIT_WS_INTS * sizeof(int)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&c->ksplit_cnt), tfmq_ctx::KSPLIT_MAX_TILES * sizeof(int)) != hipSuccess ||
      hipMemset(c->ksplit_cnt, 0, tfmq_ctx::KSPLIT_MAX_TILES * sizeof(int)) != hipSuccess) {
    if (c->ksplit_ws) (void)hipFree(c->ksplit_ws);
    if (c->ksplit_cnt) (void)hipFree(c->ksplit_cnt);
    (void)hipFree(c->pad_table);
    delete c;
    return TFMQ_ERR_HIP;
  }
  *out = c;
  return TFMQ_OK;
}

extern "C" int tfmq_destroy(tfmq_handle h) {
  if (!h) return TFMQ_ERR_ARG;
  if (h->comm) (void)tfmq_comm_destroy(h);
  for (auto g : h->graphs)
    if (g) (void)hipGraphExecDestroy(g);
  for (auto e : h->events)
    if (e) (void)hipEventDestroy(e);
  if (h->pad_table) (void)hipFree(h->pad_table);
  if (h->gemm_ws) (void)hipFree(h->gemm_ws);
  if (h->ksplit_ws) (void)hipFree(h->ksplit_ws);
  if (h->ksplit_cnt) (void)hipFree(h->ksplit_cnt);
  if (h->ksplit_ev) (void)hipEventDestroy(h->ksplit_ev);
  delete h;
  return TFMQ_OK;
}

extern "C" const char* tfmq_last_error(tfmq_handle h) { return h ? h->err.c_str() : "null handle"; }

extern "C" int tfmq_device_info(tfmq_handle h, int* cu_count, int* clock_khz, size_t* hbm_bytes) {
  if (!h) return TFMQ_ERR_ARG;
  if (cu_count) *cu_count = h->cu_count;
  if (clock_khz) *clock_khz = h->clock_khz;
  if (hbm_bytes) *hbm_bytes = h->hbm_bytes;
  return TFMQ_OK;
}

extern "C" int tfmq_graph_begin(tfmq_handle h, void* stream) {
  TFMQ_CHECK_ARG(h, h && stream, "graph_begin: capture needs a non-default stream");
  // the split-K arrival tickets are zero between launches; a launch that aborted would leave some set: clear them ahead of a capture
  TFMQ_HIP(h, hipMemsetAsync(h->ksplit_cnt, 0, tfmq_ctx::KSPLIT_MAX_TILES * sizeof(int), as_stream(stream)));
  TFMQ_HIP(h, hipStreamBeginCapture(as_stream(stream), hipStreamCaptureModeThreadLocal));
  return TFMQ_OK;
}

extern "C" int tfmq_graph_end(tfmq_handle h, void* stream, int* graph_id) {
  TFMQ_CHECK_ARG(h, h && stream && graph_id, "graph_end: bad argument");
  hipGraph_t g = nullptr;
  TFMQ_HIP(h, hipStreamEndCapture(as_stream(stream), &g));
  hipGraphExec_t ge = nullptr;
  hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  TFMQ_HIP(h, e);
  h->graphs.push_back(ge);
  *graph_id = static_cast<int>(h->graphs.size()) - 1;
  return TFMQ_OK;
}

extern "C" int tfmq_graph_launch(tfmq_handle h, int graph_id, void* stream) {
  TFMQ_CHECK_ARG(h, h && graph_id >= 0 && graph_id < static_cast<int>(h->graphs.size()) && h->graphs[graph_id],
                 "graph_launch: bad graph id");
  TFMQ_HIP(h, hipGraphLaunch(h->graphs[graph_id], as_stream(stream)));
  return TFMQ_OK;
}

extern "C" int tfmq_graph_destroy(tfmq_handle h, int graph_id) {
  TFMQ_CHECK_ARG(h, h && graph_id >= 0 && graph_id < static_cast<int>(h->graphs.size()), "graph_destroy: bad id");
  if (h->graphs[graph_id]) {
    TFMQ_HIP(h, hipGraphExecDestroy(h->graphs[graph_id]));
    h->graphs[graph_id] = nullptr;
  }
  return TFMQ_OK;
}

extern "C" int tfmq_event_create(tfmq_handle h, int* event_id) {
  TFMQ_CHECK_ARG(h, h && event_id, "event_create: bad argument");
  hipEvent_t e;
  TFMQ_HIP(h, hipEventCreate(&e));
  h->events.push_back(e);
  *event_id = static_cast<int>(h->events.size()) - 1;
  return TFMQ_OK;
}

extern "C" int tfmq_event_record(tfmq_handle h, int event_id, void* stream) {
  TFMQ_CHECK_ARG(h, h && event_id >= 0 && event_id < static_cast<int>(h->events.size()), "event_record: bad id");
  TFMQ_HIP(h, hipEventRecord(h->events[event_id], as_stream(stream)));
  return TFMQ_OK;
}

extern "C" int tfmq_event_elapsed_ms(tfmq_handle h, int start_id, int stop_id, float* ms) {
  TFMQ_CHECK_ARG(h, h && ms && start_id >= 0 && stop_id >= 0 && start_id < static_cast<int>(h->events.size()) &&
                        stop_id < static_cast<int>(h->events.size()),
                 "event_elapsed: bad id");
  TFMQ_HIP(h, hipEventSynchronize(h->events[stop_id]));
  TFMQ_HIP(h, hipEventElapsedTime(ms, h->events[start_id], h->events[stop_id]));
  return TFMQ_OK;
}

extern "C" int tfmq_stream_sync(tfmq_handle h, void* stream) {
  TFMQ_CHECK_ARG(h, h, "stream_sync: null handle");
  TFMQ_HIP(h, hipStreamSynchronize(as_stream(stream)));
  return TFMQ_OK;
}
