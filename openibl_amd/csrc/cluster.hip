// Centroid update of Lloyd's k-means for the NetVLAD centroid initialisation
// (examples/cluster.py:110-115 of the reference: scikit-learn KMeans(64, max_iter=100) on 50 000 sampled,
// L2-normalised conv5 descriptors).  The assignment step is oibl_sqdist_topk(k = 1); this file holds the
// other half: per-cluster means of the member rows.
//
// One workgroup per (cluster, 256-channel block) walks the label array once — a wave-uniform compare per
// point, the row is only read by the workgroups of the cluster it belongs to, so x is read exactly once
// overall — and accumulates in fp64 in a FIXED order (point index): the result is the correctly rounded
// fp32 mean, bit-reproducible, independent of the launch geometry.  50 000 x 512 points, 64 clusters:
// 128 workgroups x 50 000 scalar compares, ~0.3 ms.  HBM / latency bound, nothing for the matrix cores.
#include "common.h"

namespace oibl {

__global__ __launch_bounds__(256) void cluster_means_kernel(const float* __restrict__ x,
                                                            const int32_t* __restrict__ labels, int n, int d,
                                                            float* __restrict__ centers, int32_t* __restrict__ counts) {
  const int k = blockIdx.x;
  const int c = blockIdx.y * 256 + threadIdx.x;
  const bool active = c < d;
  double s = 0.0;
  int cnt = 0;
  for (int i = 0; i < n; ++i) {
    if (labels[i] == k) {          // uniform: scalar load + scalar branch
      ++cnt;
      if (active) s += (double)x[(long)i * d + c];
    }
  }
  // an empty cluster keeps its previous centre; the caller sees counts[k] == 0 and relocates it
  if (active && cnt > 0) centers[(long)k * d + c] = (float)(s / (double)cnt);
  if (blockIdx.y == 0 && threadIdx.x == 0) counts[k] = cnt;
}

}  // namespace oibl

using namespace oibl;

extern "C" int oibl_cluster_means(const float* x, const int32_t* labels, int n, int d, int num_clusters,
                                  float* centers, int32_t* counts, void* stream) {
  OIBL_REQUIRE(x && labels && centers && counts, "cluster_means: null pointer");
  OIBL_REQUIRE(n > 0 && d > 0 && num_clusters > 0 && num_clusters <= 65535, "cluster_means: bad shape");
  hipLaunchKernelGGL(cluster_means_kernel, dim3(num_clusters, (d + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, x, labels, n, d, centers, counts);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}
