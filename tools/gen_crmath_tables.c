/* tools/gen_crmath_tables.c -- prints the double-double constants of csrc/rl_crmath.hpp (atan(j/64), j = 0..64, and
 * 1/3, 1/5, 1/7, pi/2, pi) as C hex-float literals, from libquadmath (113-bit).  The header's table was pasted from
 * this program's output:   gcc -O2 tools/gen_crmath_tables.c -o /tmp/gen -lquadmath && /tmp/gen            */
#include <quadmath.h>
#include <stdio.h>
static void dd(const char* name, __float128 v, const char* tail) {
  double hi = (double)v;
  double lo = (double)(v - (__float128)hi);
  if (name) printf("%s{%a, %a}%s", name, hi, lo, tail); else printf("  {%a, %a}%s", hi, lo, tail);
}
int main(void) {
  printf("// atan(j/64), j = 0 .. 64\n");
  for (int j = 0; j <= 64; ++j) dd(NULL, atanq((__float128)j / 64), j % 2 ? ",\n" : ",");
  printf("\n");
  dd("kThird = ", (__float128)1 / 3, ";\n");
  dd("kFifth = ", (__float128)1 / 5, ";\n");
  dd("kSeventh = ", (__float128)1 / 7, ";\n");
  dd("kPio2 = ", M_PIq / 2, ";\n");
  dd("kPi = ", M_PIq, ";\n");
  /* exact-axis constants: cos / sin of fl(pi/2), fl(pi), fl(fl(pi) + fl(pi/2)) correctly rounded */
  double p2 = 0x1.921fb54442d18p+0, p = 0x1.921fb54442d18p+1, t = p + p2;
  printf("cos(fl(pi/2)) = %a   sin(fl(pi)) = %a   cos(fl(3pi/2)) = %a  (3pi/2 -> %a)\n",
         (double)cosq(p2), (double)sinq(p), (double)cosq(t), t);
  printf("sin(fl(pi/2)) = %a   cos(fl(pi)) = %a   sin(fl(3pi/2)) = %a\n", (double)sinq(p2), (double)cosq(p), (double)sinq(t));
  return 0;
}
