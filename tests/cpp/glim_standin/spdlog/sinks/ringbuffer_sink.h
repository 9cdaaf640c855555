#pragma once
#include <spdlog/spdlog.h>
