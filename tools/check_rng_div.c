// exhaustive check: for every a in [1, 2^31-2], fma-based a/2147483646 equals the IEEE quotient
#include <math.h>
#include <stdio.h>
#include <stdint.h>
int main(){
  const double b = 2147483646.0;
  const double y = 1.0 / b;          // correctly rounded reciprocal
  uint64_t bad = 0;
  for (uint32_t a = 1; a <= 2147483646u; a++) {
    double x = (double)a;
    double q0 = x * y;
    double r = fma(-b, q0, x);
    double q1 = fma(r, y, q0);
    if (q1 != x / b) { if (bad < 5) printf("mismatch a=%u %a %a\n", a, q1, x / b); bad++; }
  }
  printf("bad = %llu, y = %a\n", (unsigned long long)bad, y);
  return bad != 0;
}
