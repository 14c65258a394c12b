// TEMPORARY: engine entry points until opt.hip / mapper.hip / unet.hip land.
#include "../../include/gill_amd.h"
#include "common.h"
#define NI(name) gill_set_error(name " not implemented yet"); return -9;
extern "C" {
int gill_opt_create(gill_opt**, const gill_opt_config*, const gill_tensor*, int) { NI("gill_opt_create") }
void gill_opt_destroy(gill_opt*) {}
int gill_opt_embed(gill_opt*, const int64_t*, int, void*, void*) { NI("gill_opt_embed") }
int gill_opt_forward(gill_opt*, const void*, int, int, float*, void*) { NI("gill_opt_forward") }
int gill_opt_img_hidden(gill_opt*, const int64_t*, const int32_t*, int, int, int, void*, void*, void*) { NI("gill_opt_img_hidden") }
int gill_opt_last_logits(gill_opt*, const float*, int, int, float*, void*) { NI("gill_opt_last_logits") }
int gill_mapper_create(gill_mapper**, const gill_mapper_config*, const gill_tensor*, int) { NI("gill_mapper_create") }
void gill_mapper_destroy(gill_mapper*) {}
int gill_mapper_forward(gill_mapper*, const void*, const void*, int, int, float*, void*) { NI("gill_mapper_forward") }
int gill_unet_create(gill_unet**, const gill_unet_config*, const gill_tensor*, int) { NI("gill_unet_create") }
void gill_unet_destroy(gill_unet*) {}
int gill_unet_forward(gill_unet*, const float*, const float*, const void*, int, float*, void*) { NI("gill_unet_forward") }
int gill_sd_denoise(gill_unet*, const void*, const void*, const float*, int, int, float, float*, void*) { NI("gill_sd_denoise") }
int gill_pndm_schedule(int, int32_t*, double*) { NI("gill_pndm_schedule") }
}
