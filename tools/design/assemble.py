"""DESIGN.md = the section files beside this script + the figures of profiles/r06_bench_line.json and profiles/r06_parity_report.json (python tools/design/assemble.py)."""
import json, os, sys, re
D=os.path.dirname(os.path.abspath(__file__))+'/'
R=os.path.dirname(os.path.dirname(D.rstrip('/')))+'/'
bl=json.loads(open(R+'profiles/r06_bench_line.json').read().strip().splitlines()[-1])
pr=json.load(open(R+'profiles/r06_parity_report.json'))
kc=bl['kernel_classes']
def cls(k): v=kc[k]; return f"{v['ms']:.1f} ms at {v['TFLOPs']:.0f} TFLOP/s ({v['frac']:.2f})" if v['TFLOPs']>1 else f"{v['ms']:.1f} ms at {v['GBs']:.0f} GB/s ({v['frac']:.2f} of HBM)"
s0=open(D+'s0.md').read()
da=bl['daam_accumulate']
rep={'@MS@':f"{bl['ms_per_step']:.1f}",'@IPS@':f"{bl['value']:.2f}",'@E2E@':f"{bl['roofline']['end_to_end_frac']:.3f}",'@E2EX@':f"{bl['roofline']['end_to_end_frac_executed']:.3f}",
     '@CONV@':cls('igemm_conv3x3'),'@LIN@':cls('igemm_linear_1x1'),'@SELF@':cls('attn_self_flash'),'@CROSS@':cls('attn_cross_daam'),'@GN@':cls('groupnorm'),
     '@DDELTA@':f"{da['delta_ms']:.2f}",'@DGBS@':f"{da.get('accumulate_GBs_on_delta',0)/1000:.1f}"}
for k,v in rep.items(): s0=s0.replace(k,v)
s2=open(D+'s2.md').read()
def g(i,*ks): return ", ".join(f"{k.replace('_rms_rel','').replace('norm_map_','nm ').replace('_255','')} {pr[i][k]:.4g}" for k in ks if k in pr.get(i,{}))
c2=pr['config2_512px_50_steps_end_to_end']; c1=pr['config1_256px_10_steps_end_to_end']
rep2={'@C2@':f"latents {c2['latents_rms_rel']:.4f} / {c2['psnr_db']:.1f} dB / heat map {100*c2['heat_map_rel']:.2f} % / max {c2['norm_map_max_255']:.2f}, p99.9 {c2.get('norm_map_p999_255',0):.2f}, mean {c2['norm_map_mean_255']:.2f}",
 '@C2F@':f"rms rel {pr['config2_forward_512px_batch4']['rms_rel']:.4f} / heat map {100*pr['config2_forward_512px_batch4']['heat_map_rel']:.2f} %",
 '@C1@':f"latents {c1['latents_rms_rel']:.4f} / {c1['psnr_db']:.1f} dB / heat map {100*c1['heat_map_rel']:.2f} % / max {c1['norm_map_max_255']:.2f}, p99.9 {c1['norm_map_p999_255']:.2f}, mean {c1['norm_map_mean_255']:.2f}",
 '@C3@':f"worst image {pr['config3_share_forward_512px_unet_batch16']['worst_image_rms_rel']:.4f}; moments {pr['config3_share_vae_encode_512px_batch8']['moments_mean_rms_rel_worst']:.4f} / {pr['config3_share_vae_encode_512px_batch8']['moments_logvar_rms_rel_worst']:.4f}; latents {pr['config3_vae_encode_img2img_512px']['latents_rms_rel']:.4f} / {pr['config3_vae_encode_img2img_512px']['psnr_db']:.1f} dB",
 '@C4@':f"{pr['config4_learned_token_heat_maps_png']['worst_abs_255']:.0f} / 255",
 '@C5@':f"worst image {pr['config5_share_forward_768px_unet_batch8']['worst_image_rms_rel']:.4f}; {pr['config5_vae_decode_768px']['rms_rel']:.4f} / {pr['config5_vae_decode_768px']['psnr_db']:.1f} dB; latents {pr['config5_vpred_3_steps_768px']['latents_rms_rel']:.4f}",
 '@ODD@':"; ".join(f"{k.split('[')[1][:-1]}: {pr[k]['latents_rms_rel']:.3f}, vs oracle {pr[k]['merged_vs_oracle']:.3f} / {pr[k]['unmerged_vs_oracle']:.3f}" for k in sorted(pr) if k.startswith('odd_size')),
 '@GOLD@':"; ".join(f"{k.split('[')[1][:-1]}: {pr[k]['out_max_rel']:.4f} / {pr[k]['map_max_abs']:.1e}" for k in sorted(pr) if k.startswith('golden_')),
}
for k,v in rep2.items(): s2=s2.replace(k,v)
s3=open(D+'s3.md').read().rstrip('\n')+'''
* Round 6: the row-panel kernels' output rows pass through a wave-private LDS transpose (5 KiB per wave) so that consecutive lanes store consecutive 16-byte pieces of whole 160-byte row segments; `ff_fused`
  keeps three chunks of GEGLU epilogue constants (6 KiB) and `attn_chain` norm2's gamma / beta (2 C floats) in LDS.

'''
s4=open(D+'s4.md').read()
s4=s4.replace("\n\nKernel notes, measurements and every measured-and-rejected variant: Appendix A (round by round; the numbers there belong to the round that wrote them -- the current ones are in section 0).",
 "\n| **round 6** `qkv_chain2_kernel<C, MH>` (tblock.hip) | the block head on the schedule its stamps asked for, 64-row panels as two co-resident four-wave workgroups per CU (section 4.2); bit-identical to `qkv_chain_kernel` | HBM writes (84 MB per launch) | as `qkv_chain_kernel` |\n")
out=s0+"\n"+open(D+'s1.md').read()+s2+"\n"+s3+s4+"\n"+open(D+'s4b.md').read()+"\n"+open(D+'s5.md').read()+"\n"+open(D+'s6.md').read()+open(D+'s7.md').read()+open(D+'s8.md').read()
open(R+'DESIGN.md','w').write(out)
print(len(out.encode()))
