// Run-time options of libhx.
//
// Every tuning / experiment switch of the library used to be an environment variable read ONCE into a function-local static:
// a test (or a driver) that set the variable after the first launch silently changed nothing.  Options now live in one
// registry: hx_set_option(name, value) overrides, hx_get_option reads back, and a switch that was never set falls back to the
// environment variable HX_<NAME IN UPPER CASE> (so existing scripts keep working).  Call sites cache their value together
// with the registry's generation number, so a lookup costs one relaxed atomic load per launch and a change takes effect at
// the next call.  No reference counterpart: PETSc's options database (-st_type, -eps_tol ...; the reference sets its solver
// options in code, eigensolvers.py:43-60) plays this role there.
#include <atomic>
#include <cctype>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

#include "hx_common.h"

namespace {
std::mutex g_mu;
std::map<std::string, std::string> g_opts;
}  // namespace
std::atomic<int> hx_option_generation{0};

static bool lookup(const char* name, std::string& out) {
  {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_opts.find(name);
    if (it != g_opts.end()) { out = it->second; return true; }
  }
  std::string env = "HX_";
  for (const char* p = name; *p; ++p) env.push_back((char)toupper((unsigned char)*p));
  const char* e = getenv(env.c_str());
  if (e) { out = e; return true; }
  return false;
}

long hx_option_lookup_int(const char* name, long dflt) {
  std::string v;
  if (!lookup(name, v) || v.empty()) return dflt;
  char* end = nullptr;
  const long r = strtol(v.c_str(), &end, 10);
  return end == v.c_str() ? dflt : r;
}
double hx_option_lookup_double(const char* name, double dflt) {
  std::string v;
  if (!lookup(name, v) || v.empty()) return dflt;
  char* end = nullptr;
  const double r = strtod(v.c_str(), &end);
  return end == v.c_str() ? dflt : r;
}
bool hx_option_lookup_is(const char* name, const char* value) {
  std::string v;
  return lookup(name, v) && v == value;
}
bool hx_option_lookup_set(const char* name) {
  std::string v;
  return lookup(name, v);
}

// value == NULL (or ""): remove the override (the environment variable, then the built-in default, apply again)
extern "C" int hx_set_option(const char* name, const char* value) {
  HX_REQUIRE(name && *name, "option name");
  {
    std::lock_guard<std::mutex> lock(g_mu);
    if (value && *value) g_opts[name] = value;
    else g_opts.erase(name);
  }
  hx_option_generation.fetch_add(1);
  return HX_OK;
}
// the value in force (override, else environment) as text; returns HX_ERR_ARG if neither is set or the buffer is too small
extern "C" int hx_get_option(const char* name, char* out, int out_len) {
  HX_REQUIRE(name && out && out_len > 0, "bad argument");
  std::string v;
  if (!lookup(name, v)) return hx_fail(HX_ERR_ARG, "hx_get_option", "option not set");
  HX_REQUIRE((int)v.size() < out_len, "buffer too small");
  memcpy(out, v.c_str(), v.size() + 1);
  return HX_OK;
}
