// debugging shim: lets the device header compile as host C++ (scratch only)
#pragma once
#include <cmath>
#include <cstring>
#include <cstdint>
#define __device__
#define __host__
#define __forceinline__ inline
#define __global__
static inline double __longlong_as_double(long long v){double d; memcpy(&d,&v,8); return d;}
static inline long long __double_as_longlong(double d){long long v; memcpy(&v,&d,8); return v;}
using std::signbit; using std::fmax; using std::fabs; using std::sqrt;
