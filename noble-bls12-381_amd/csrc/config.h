// config.h -- the ONE place where the library reads its environment switches (round 4; 30 getenv sites before).  Every switch is read once per process, on first
// use, through env_long / env_set; the value then in force is remembered, and env_describe() (C ABI: nbls_config_describe) lists every switch asked for so far
// with its value and where it came from -- so that an A/B script can print what each side actually ran with, and a switch set after its first use (which has no
// effect) shows up as such.  Tuning values that may change at run time go through nbls_set_tuning (include/nbls.h), not through the environment.
#pragma once
#include <string>
namespace nbls {
long env_long(const char* name, long dflt);   // integer value of the variable, or dflt when it is unset or empty
bool env_set(const char* name);               // the variable is defined (whatever its value)
std::string env_describe();                   // "NAME=value (env|default) ..." of every switch read so far, in order of first use
}  // namespace nbls
