// placeholder until the GSO device path lands; keeps the context teardown symmetric
#include "../../include/fplll_hip.h"
extern "C" void fphip_gso_release_all(fphip_ctx *ctx) { (void)ctx; }
