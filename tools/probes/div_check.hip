// Is  a / b  ==  the five-instruction sequence with a correctly rounded reciprocal, bit for bit?
//     y = 1 / b (true division, once per divisor);  q0 = a y;  r0 = fma(-b, q0, a);  q1 = fma(r0, y, q0);
//     r1 = fma(-b, q1, a);  q = fma(r1, y, q1)
// (Markstein: q1 is within half an ulp + of a/b, hence faithful, and the second correction with y = RN(1/b) rounds
// correctly; no underflow / overflow assumed -- the caller guards the exponent ranges.)  The fused EEG kernel divides
// 4 x 1,081 numbers by 47 standard deviations per window (np.corrcoef's c / s_i / s_j, both ways round).
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -Wno-unused-value tools/probes/div_check.hip -o tools/probes/div_check && tools/probes/div_check
// Random operands over 40 binades, operands with few significant bits, quotients next to rounding boundaries
// (a = q_mid * b rounded, q_mid a midpoint between two doubles), divisors with all-ones significands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ double fast_div(double a, double b, double y)
{
    const double q0 = a * y;
    const double r0 = fma(-b, q0, a);
    const double q1 = fma(r0, y, q0);
    const double r1 = fma(-b, q1, a);
    return fma(r1, y, q1);
}
__device__ __forceinline__ uint64_t splitmix(uint64_t& s)
{
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double from_bits(uint64_t m, int e, bool neg)
{
    const uint64_t bits = ((uint64_t)(neg ? 1 : 0) << 63) | ((uint64_t)(1023 + e) << 52) | (m & 0xfffffffffffffull);
    return __longlong_as_double((long long)bits);
}

__global__ void __launch_bounds__(256) k_check(uint64_t seed, int iters, int mode, unsigned long long* bad, double* ex)
{
    uint64_t s = seed + 0x1234567ull * (blockIdx.x * 256ull + threadIdx.x);
    unsigned long long nb = 0;
    for (int it = 0; it < iters; ++it) {
        const uint64_t r1 = splitmix(s), r2 = splitmix(s), r3 = splitmix(s);
        const int ea = (int)(r3 & 63) - 32, eb = (int)((r3 >> 6) & 63) - 32;
        uint64_t ma = r1, mb = r2;
        if (mode == 1) { ma &= ~((1ull << (int)((r3 >> 12) % 50)) - 1ull); mb &= ~((1ull << (int)((r3 >> 18) % 50)) - 1ull); }   // few bits
        if (mode == 3) mb = 0xfffffffffffffull ^ ((r3 >> 24) & 7);                                                            // ~all ones
        double b = from_bits(mb, eb, false), a = from_bits(ma, ea, (r3 >> 40) & 1);
        if (mode == 2) {                           // a such that a / b lies next to a midpoint: a = RN((q + ulp/2 +- tiny) b)
            const double q = from_bits(ma, ea - eb, false);
            const double qn = __longlong_as_double(__double_as_longlong(q) + 1);
            const double mid_lo = q, mid_hi = qn;                       // (q + qn) / 2 is not representable: aim a at it via fma
            a = fma(0.5 * (mid_hi - mid_lo), b, q * b);                 // ~ (q + ulp/2) b, rounded
            if ((r3 >> 41) & 1) a = __longlong_as_double(__double_as_longlong(a) + (long long)((r3 >> 42) & 3) - 1);
        }
        const double y = 1.0 / b;
        const double q_true = a / b, q_fast = fast_div(a, b, y);
        if (__double_as_longlong(q_true) != __double_as_longlong(q_fast)) {
            if (nb == 0 && atomicAdd(bad + 1, 1ull) == 0ull) { ex[0] = a; ex[1] = b; ex[2] = q_true; ex[3] = q_fast; }
            ++nb;
        }
    }
    if (nb) atomicAdd(bad, nb);
}

int main()
{
    unsigned long long* bad; double* ex;
    hipMalloc(&bad, 16); hipMalloc(&ex, 32);
    const char* names[4] = {"random operands", "operands with few significant bits", "quotients next to a rounding boundary", "divisors with (nearly) all-ones significands"};
    for (int mode = 0; mode < 4; ++mode) {
        hipMemset(bad, 0, 16);
        const int blocks = 4096, iters = mode == 0 ? 8192 : 2048;
        for (int rep = 0; rep < 4; ++rep) k_check<<<blocks, 256>>>(0xabcdef12345ull + 977ull * rep + 131ull * mode, iters, mode, bad, ex);
        hipDeviceSynchronize();
        unsigned long long hb[2]; double he[4];
        hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost); hipMemcpy(he, ex, 32, hipMemcpyDeviceToHost);
        printf("%-48s %.3g cases, %llu mismatches", names[mode], 4.0 * blocks * 256.0 * iters, hb[0]);
        if (hb[0]) printf("  e.g. a=%a b=%a a/b=%a fast=%a", he[0], he[1], he[2], he[3]);
        printf("\n");
    }
    return 0;
}
