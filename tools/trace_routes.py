#!/usr/bin/env python3
"""Kernel-development aid: LRCN_TRACE_ROUTES=1 tools/trace_routes.py -> the kernel family every contraction of one 32-row lossgradient
(beside-the-VGG dispatch: weights loaded, grid cap 160) takes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lrcn_amd
from lrcn_amd import lrcn as L
E=H=1000; V,B,T=10640,32,11
ctx=L.Context(E,H,H,V,max_B=B,max_T=T,lstm_dtype=lrcn_amd.LRCN_BF16,vgg_dtype=lrcn_amd.LRCN_BF16,max_images=1)
L.vgg_load(ctx,*L.synthetic_vgg_weights(seed=1)); L.vgg_set_wg_cap(ctx,160)
param=L.initweights(ctx,seed=1)
rng=np.random.default_rng(0)
feats=L.to_jl((rng.standard_normal((B,4096))*0.01).astype(np.float32)); tok=rng.integers(3,V,size=(T,B)).astype(np.int32)
L.lossgradient(ctx,param,feats,tok,norm_B=256,pdrop=0.4,seed=1)
