// log.h — the reference's logging convention (src/utils.cpp:15-29): mutex-guarded stdout / stderr lines.
#pragma once
#include <string>
void printMessage(const std::string &message);
void printError(const std::string &message);
namespace csvhost { void set_quiet(bool quiet); }
