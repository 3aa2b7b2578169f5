#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdint.h>
static uint32_t f2u(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static float u2f(uint32_t u){float f;memcpy(&f,&u,4);return f;}
static float advance(float x, const float s, uint32_t n, int *iters) {
    while (n != 0u) {
        (*iters)++;
        const float x1 = x + s;
        if (--n == 0u) return x1;
        const float x1b = x1 + s;
        if (--n == 0u) return x1b;
        const float x2 = x1b + s;
        --n;
        x = x2;
        const uint32_t e1 = f2u(x1b) & 0x7f800000u, e2 = f2u(x2) & 0x7f800000u;
        const float inc = x2 - x1b;
        if (n == 0u || e1 != e2 || (f2u(x1) & 0x7f800000u) != e1 || e2 == 0u || e2 == 0x7f800000u || inc == 0.0f) { if (inc == 0.0f && e1 == e2) return x2; continue; }
        const float B = u2f(e2), ax = fabsf(x2), ai = fabsf(inc);
        const int away = (inc > 0.0f) == (x2 > 0.0f);
        const float room = away ? fmaf(2.0f, B, -ax) : ax - B;
        const float kf = floorf(room / ai) - 2.0f;
        if (!(kf >= 1.0f)) continue;
        const uint32_t k = (uint32_t)fminf(kf, (float)n);
        x = fmaf((float)k, inc, x2);
        n -= k;
    }
    return x;
}
int main(){
    srand(12345);
    long bad=0, tot=0; int maxit=0; double maxerr=0; long sumit=0;
    for (int c=0;c<2000000;c++){
        float x = (float)rand()/RAND_MAX*1.4f-0.2f;
        float s = ((float)rand()/RAND_MAX*0.01f+1e-5f) * ((rand()&1)?1.f:-1.f);
        if (c%7==0) x = (rand()&1)? 0.0f : 1.0f;
        uint32_t n = rand()%700;
        volatile float r = x; for (uint32_t i=0;i<n;i++) r = r + s;
        int it=0; float a = advance(x,s,n,&it);
        tot++; sumit+=it; if (it>maxit) maxit=it;
        if (a != r){ bad++; double e=fabs((double)a-(double)r); if(e>maxerr)maxerr=e; if (bad<6) printf("x=%a s=%a n=%u ref=%a got=%a\n",x,s,n,(float)r,a);}
    }
    printf("bad %ld of %ld, max err %g, max iters %d, mean iters %.2f\n",bad,tot,maxerr,maxit,(double)sumit/tot);
    return 0;
}
