// Tuning build of the aggregation kernel (not part of the product library).
#define BGNN_TUNING 1
#include "../bridged_gnn_amd/csrc/bgnn_aggregate.hip"
