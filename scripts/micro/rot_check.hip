// device vs host bits of the sub-expressions of epnp::jacobi_rotation (hipcc -O3 -ffp-contract=off ...; run on the GPU box)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
struct Out { double h, g, q, sq, r, c, s; };
__host__ __device__ inline Out rot(double app, double aqq, double apq) {
  Out o;
  const double d = aqq - app, b = 2.0 * apq;
  const double ad = fabs(d), ab = fabs(b);
  o.h = sqrt(d * d + b * b);
  o.g = ad + o.h;
  o.q = (2.0 * o.h) * o.g;
  o.sq = sqrt(o.q);
  o.r = 1.0 / sqrt((2.0 * o.h) * o.g);
  o.c = o.g * o.r;
  o.s = ab * o.r;
  return o;
}
__global__ void k(const double* in, Out* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = rot(in[3 * i], in[3 * i + 1], in[3 * i + 2]);
}
int main() {
  const int n = 1 << 20;
  std::vector<double> in(3 * n);
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> u(-1, 1);
  for (int i = 0; i < n; i++) {
    double sc = std::pow(10.0, 6 * u(g));
    in[3 * i] = sc * u(g), in[3 * i + 1] = sc * u(g), in[3 * i + 2] = sc * u(g) * std::pow(10.0, -8 * std::fabs(u(g)));
  }
  double* din; Out* dout;
  hipMalloc(&din, sizeof(double) * 3 * n); hipMalloc(&dout, sizeof(Out) * n);
  hipMemcpy(din, in.data(), sizeof(double) * 3 * n, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(din, dout, n);
  std::vector<Out> o(n);
  hipMemcpy(o.data(), dout, sizeof(Out) * n, hipMemcpyDeviceToHost);
  long bad[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; i++) {
    Out h = rot(in[3 * i], in[3 * i + 1], in[3 * i + 2]);
    const double* a = &h.h; const double* b = &o[i].h;
    for (int j = 0; j < 7; j++) bad[j] += std::memcmp(a + j, b + j, 8) != 0;
  }
  printf("mismatches of %d: h %ld g %ld q %ld sqrt(q) %ld 1/sqrt(q) %ld c %ld s %ld\n", n, bad[0], bad[1], bad[2], bad[3], bad[4], bad[5], bad[6]);
  return 0;
}
