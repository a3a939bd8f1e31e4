// One kernel dispatched by writing the AQL packet ourselves (HSA user-mode queue, no HIP launch path), next to
// tools/launch_probe.hip: host cost of the dispatch and dispatch -> the kernel's word visible in pinned memory.
//   hipcc --genco --offload-arch=gfx950 -o k.hsaco k.hip   (extern "C" __global__ void k_big(Big))
//   hipcc -O2 -o aql_probe tools/aql_probe.cpp -lhsa-runtime64 ; ./aql_probe k.hsaco
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <vector>
struct Big { unsigned long long *out; long long seq; char pad[1088]; };
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define HS(x) do { hsa_status_t e_ = (x); if (e_ != HSA_STATUS_SUCCESS) { const char *m = ""; hsa_status_string(e_, &m); printf("%s: %s\n", #x, m); return 1; } } while (0)
static hsa_agent_t g_gpu, g_cpu; static bool have_gpu = false, have_cpu = false;
static hsa_amd_memory_pool_t g_kpool; static bool have_kpool = false;
static hsa_status_t on_agent(hsa_agent_t a, void *) {
  hsa_device_type_t t;
  hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && !have_gpu) { g_gpu = a; have_gpu = true; }
  if (t == HSA_DEVICE_TYPE_CPU && !have_cpu) { g_cpu = a; have_cpu = true; }
  return HSA_STATUS_SUCCESS;
}
static hsa_status_t on_pool(hsa_amd_memory_pool_t p, void *) {
  hsa_amd_segment_t seg;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
  uint32_t flags = 0;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
  if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !have_kpool) { g_kpool = p; have_kpool = true; }
  return HSA_STATUS_SUCCESS;
}
int main(int argc, char **argv) {
  if (argc < 2) return 2;
  unsigned long long *host = nullptr;
  if (hipHostMalloc(&host, 64, hipHostMallocMapped) != hipSuccess) return 3;
  HS(hsa_init());
  HS(hsa_iterate_agents(on_agent, nullptr));
  if (!have_gpu || !have_cpu) { printf("no agents\n"); return 4; }
  HS(hsa_amd_agent_iterate_memory_pools(g_cpu, on_pool, nullptr));
  if (!have_kpool) { printf("no kernarg pool\n"); return 5; }
  const int fd = open(argv[1], O_RDONLY);
  if (fd < 0) { printf("cannot open %s\n", argv[1]); return 6; }
  hsa_code_object_reader_t reader;
  HS(hsa_code_object_reader_create_from_file(fd, &reader));
  hsa_executable_t exe;
  HS(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
  HS(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
  HS(hsa_executable_freeze(exe, nullptr));
  hsa_executable_symbol_t sym;
  HS(hsa_executable_get_symbol_by_name(exe, "k_big.kd", &g_gpu, &sym));
  uint64_t kobj = 0; uint32_t ksize = 0, gseg = 0, pseg = 0;
  HS(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kobj));
  HS(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &ksize));
  HS(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &gseg));
  HS(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &pseg));
  printf("kernel object %llx kernarg %u group %u private %u\n", (unsigned long long)kobj, ksize, gseg, pseg);
  hsa_queue_t *q = nullptr;
  HS(hsa_queue_create(g_gpu, 256, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
  const int kSlots = 16;
  const size_t slot = (ksize + 255) & ~size_t(255);
  char *kargs = nullptr;
  HS(hsa_amd_memory_pool_allocate(g_kpool, slot * kSlots, 0, reinterpret_cast<void **>(&kargs)));
  HS(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, kargs));
  Big b{};
  b.out = host;
  const int N = 3000;
  std::vector<double> call, seen;
  for (int i = 0; i < N + 200; ++i) {
    b.seq = i + 1;
    *host = 0;
    const double t0 = now();
    char *ka = kargs + slot * (i % kSlots);
    std::memcpy(ka, &b, sizeof(b));
    const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
    hsa_kernel_dispatch_packet_t *p = reinterpret_cast<hsa_kernel_dispatch_packet_t *>(q->base_address) + (idx & (q->size - 1));
    p->setup = 1;  // one dimension
    p->workgroup_size_x = 1024; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
    p->grid_size_x = 256 * 1024; p->grid_size_y = 1; p->grid_size_z = 1;
    p->private_segment_size = pseg;
    p->group_segment_size = gseg;
    p->kernel_object = kobj;
    p->kernarg_address = ka;
    p->reserved2 = 0;
    p->completion_signal.handle = 0;
    const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                            (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                            (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
    __atomic_store_n(reinterpret_cast<uint16_t *>(p), header, __ATOMIC_RELEASE);
    hsa_signal_store_screlease(q->doorbell_signal, static_cast<hsa_signal_value_t>(idx));
    const double t1 = now();
    while (*reinterpret_cast<volatile unsigned long long *>(host) != static_cast<unsigned long long>(b.seq)) {}
    const double t2 = now();
    if (i >= 200) { call.push_back(t1 - t0); seen.push_back(t2 - t0); }
  }
  std::sort(call.begin(), call.end());
  std::sort(seen.begin(), seen.end());
  printf("%-32s call p50 %.2f us  p10 %.2f | call -> word seen p50 %.2f us p10 %.2f\n", "AQL packet + doorbell", call[N / 2], call[N / 10],
         seen[N / 2], seen[N / 10]);
  // the same with agent-scope fences (no system-scope cache actions around the kernel)
  call.clear(); seen.clear();
  for (int i = 0; i < N + 200; ++i) {
    b.seq = 100000 + i;
    *host = 0;
    const double t0 = now();
    char *ka = kargs + slot * (i % kSlots);
    std::memcpy(ka, &b, sizeof(b));
    const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
    hsa_kernel_dispatch_packet_t *p = reinterpret_cast<hsa_kernel_dispatch_packet_t *>(q->base_address) + (idx & (q->size - 1));
    p->setup = 1;
    p->workgroup_size_x = 1024; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
    p->grid_size_x = 256 * 1024; p->grid_size_y = 1; p->grid_size_z = 1;
    p->private_segment_size = pseg;
    p->group_segment_size = gseg;
    p->kernel_object = kobj;
    p->kernarg_address = ka;
    p->reserved2 = 0;
    p->completion_signal.handle = 0;
    const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                            (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                            (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
    __atomic_store_n(reinterpret_cast<uint16_t *>(p), header, __ATOMIC_RELEASE);
    hsa_signal_store_screlease(q->doorbell_signal, static_cast<hsa_signal_value_t>(idx));
    const double t1 = now();
    while (*reinterpret_cast<volatile unsigned long long *>(host) != static_cast<unsigned long long>(b.seq)) {}
    const double t2 = now();
    if (i >= 200) { call.push_back(t1 - t0); seen.push_back(t2 - t0); }
  }
  std::sort(call.begin(), call.end());
  std::sort(seen.begin(), seen.end());
  printf("%-32s call p50 %.2f us  p10 %.2f | call -> word seen p50 %.2f us p10 %.2f\n", "AQL, agent-scope fences", call[N / 2], call[N / 10],
         seen[N / 2], seen[N / 10]);
  hsa_queue_destroy(q);
  return 0;
}
