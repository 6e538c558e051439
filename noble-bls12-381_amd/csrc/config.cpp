// config.cpp -- see config.h
#include "config.h"
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>
namespace nbls {
namespace {
struct Entry { long value; bool from_env; bool flag_only; };
// function-local statics: switches are read during the static initialisation of other translation units (programs.cpp: lanes per item of the program families)
struct State { std::mutex mu; std::map<std::string, Entry> seen; std::vector<std::string> order; };
State& state() { static State* s = new State(); return *s; }   // never destroyed: read again from static destructors at exit
}  // namespace
#define g_mu state().mu
#define g_seen state().seen
#define g_order state().order
long env_long(const char* name, long dflt) {
  std::lock_guard<std::mutex> g(g_mu);
  auto it = g_seen.find(name);
  if (it != g_seen.end() && !it->second.flag_only) return it->second.value;
  const char* v = getenv(name);
  Entry e{dflt, false, false};
  if (v && *v) { e.value = atol(v); e.from_env = true; }
  if (it == g_seen.end()) g_order.push_back(name);
  g_seen[name] = e;
  return e.value;
}
bool env_set(const char* name) {
  std::lock_guard<std::mutex> g(g_mu);
  auto it = g_seen.find(name);
  if (it != g_seen.end()) return it->second.from_env;
  const bool on = getenv(name) != nullptr;
  g_seen[name] = Entry{on ? 1 : 0, on, true};
  g_order.push_back(name);
  return on;
}
std::string env_describe() {
  std::lock_guard<std::mutex> g(g_mu);
  std::string s;
  for (auto& n : g_order) { const Entry& e = g_seen[n]; s += (s.empty() ? "" : " ") + n + "=" + std::to_string(e.value) + (e.from_env ? "(env)" : "(default)"); }
  return s;
}
}  // namespace nbls
