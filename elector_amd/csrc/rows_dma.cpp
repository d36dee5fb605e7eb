// elector_amd/csrc/rows_dma.cpp -- the merged rows' way to the host on the GPU's DMA engine, asked for from the HSA runtime
// directly (hsa_amd_memory_async_copy: source agent the GPU, destination agent a CPU -> the SDMA engine).
//
// Why not hipMemcpyAsync: the HIP runtime decides per call whether a device-to-pinned-host copy goes to the DMA engine or
// runs as a blit KERNEL on the compute units, and in the pipeline of bench.py / getPOA it chose the kernel for four rows
// copies in five -- rocprofv3 over the default command (round 5, profiles/r05_rows_copy_engine.txt): 46 of 56 rows copies
// were __amd_rocclr_copyBuffer launches, 6-7 ms of 512-thread workgroups each, beside the alignment kernels; one copy at a
// time per device and a copy stream without event records changed nothing or made it all of them.  The HSA call takes no
// such decision.  The destination is page-locked memory the HIP runtime allocated or registered (known to HSA, which HIP
// sits on); the source a device allocation; the caller has waited for the kernels that wrote it.
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace elector {

namespace {

struct Agents {
  bool tried = false, ok = false;
  std::vector<hsa_agent_t> gpus;
  std::vector<uint32_t> gpu_bdf;           // (domain << 16) | bdfid
  hsa_agent_t cpu{};
  bool have_cpu = false;
};
Agents g_agents;
std::mutex g_mu;

hsa_status_t on_agent(hsa_agent_t ag, void *data)
{
  Agents *a = static_cast<Agents *>(data);
  hsa_device_type_t type;
  if (hsa_agent_get_info(ag, HSA_AGENT_INFO_DEVICE, &type) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
  if (type == HSA_DEVICE_TYPE_CPU) {
    if (!a->have_cpu) { a->cpu = ag; a->have_cpu = true; }
  } else if (type == HSA_DEVICE_TYPE_GPU) {
    uint32_t bdf = 0, dom = 0;
    (void)hsa_agent_get_info(ag, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
    (void)hsa_agent_get_info(ag, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &dom);
    a->gpus.push_back(ag);
    a->gpu_bdf.push_back((dom << 16) | (bdf & 0xFFFFu));
  }
  return HSA_STATUS_SUCCESS;
}

// the HSA agent of a HIP device, by PCI address; false when there is none (the caller then copies through HIP).  The first
// call per device asks the HIP runtime for the address; later ones only read the table (a stream's host function, which may
// not call into HIP, can therefore start a copy once rows_dma_ready() has been through here on the caller's thread).
constexpr int kMaxDev = 64;
int g_dev_state[kMaxDev];                  // 0 unknown, 1 agent known, -1 none
hsa_agent_t g_dev_agent[kMaxDev];

bool agent_of(int hip_device, hsa_agent_t *gpu, hsa_agent_t *cpu)
{
  if (hip_device < 0 || hip_device >= kMaxDev) return false;
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_dev_state[hip_device] == 0) {
    g_dev_state[hip_device] = -1;
    if (!g_agents.tried) {
      g_agents.tried = true;
      if (hsa_init() == HSA_STATUS_SUCCESS && hsa_iterate_agents(on_agent, &g_agents) == HSA_STATUS_SUCCESS)
        g_agents.ok = g_agents.have_cpu && !g_agents.gpus.empty();
    }
    char id[64] = {0};
    unsigned dom = 0, bus = 0, dev = 0, fn = 0;
    if (g_agents.ok && hipDeviceGetPCIBusId(id, (int)sizeof id, hip_device) == hipSuccess &&
        std::sscanf(id, "%x:%x:%x.%x", &dom, &bus, &dev, &fn) == 4) {
      const uint32_t want = (dom << 16) | ((bus & 0xFFu) << 8) | ((dev & 0x1Fu) << 3) | (fn & 7u);
      for (size_t k = 0; k < g_agents.gpus.size(); ++k)
        if (g_agents.gpu_bdf[k] == want) { g_dev_agent[hip_device] = g_agents.gpus[k]; g_dev_state[hip_device] = 1; break; }
    }
  }
  if (g_dev_state[hip_device] != 1) return false;
  *gpu = g_dev_agent[hip_device];
  *cpu = g_agents.cpu;
  return true;
}

bool dma_off()
{
  static const bool off = std::getenv("ELECTOR_ROWS_HIP_COPY") && std::atoi(std::getenv("ELECTOR_ROWS_HIP_COPY")) != 0;
  return off;
}

}  // namespace

// Can copies of this device go the HSA way?  Called on the caller's thread (it may ask the HIP runtime for the device's PCI
// address); creates the slot's signal, so that rows_dma_start() behind it makes HSA calls only.
bool rows_dma_ready(int hip_device, uint64_t *sig)
{
  if (dma_off()) return false;
  hsa_agent_t gpu, cpu;
  if (!agent_of(hip_device, &gpu, &cpu)) return false;
  if (!*sig) {
    hsa_signal_t s;
    if (hsa_signal_create(0, 0, nullptr, &s) != HSA_STATUS_SUCCESS) return false;
    *sig = s.handle;
  }
  return true;
}

// Start the copy of n bytes from device memory to page-locked host memory on the DMA engine.  *sig: the slot's signal
// (created here at the first use, value 0 = none).  -> 0, or non-zero when the HSA way is not available (nothing started).
int rows_dma_start(int hip_device, void *dst, const void *src, size_t n, uint64_t *sig)
{
  if (dma_off()) return 1;
  hsa_agent_t gpu, cpu;
  if (!agent_of(hip_device, &gpu, &cpu)) return 1;
  hsa_signal_t s;
  s.handle = *sig;
  if (!s.handle) {
    if (hsa_signal_create(0, 0, nullptr, &s) != HSA_STATUS_SUCCESS) return 1;
    *sig = s.handle;
  }
  hsa_signal_store_relaxed(s, 1);
  if (hsa_amd_memory_async_copy(dst, cpu, src, gpu, n, 0, nullptr, s) != HSA_STATUS_SUCCESS) {
    hsa_signal_store_relaxed(s, 0);
    return 1;
  }
  return 0;
}

// -> 0 when the copy has arrived, non-zero on an error
int rows_dma_wait(uint64_t sig)
{
  hsa_signal_t s;
  s.handle = sig;
  if (!s.handle) return 0;
  const hsa_signal_value_t v = hsa_signal_wait_scacquire(s, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
  return v < 0 ? 1 : 0;
}

void rows_dma_release(uint64_t *sig)
{
  hsa_signal_t s;
  s.handle = *sig;
  if (s.handle) (void)hsa_signal_destroy(s);
  *sig = 0;
}

}  // namespace elector
