// index_factory for the two descriptions the hot path needs (Auncel/AutoTune.cpp:800-831): "IVF<n>,Flat" and "Flat"
#pragma once
#include "Index.h"
#include <unordered_map>
#include <vector>

namespace faiss {
Index* index_factory(int d, const char* description, MetricType metric = METRIC_L2);
}
