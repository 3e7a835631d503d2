"""CPU restatement of the reference's layer graphs as plain functions.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
/root/reference/cnn_wrapper/network.py (op semantics) and
/root/reference/cnn_wrapper/atvsnet.py (graphs).  ``W`` maps TF variable
names to float32 torch tensors in TF layout.
All tensors channel-last float32 on the CPU; every BN uses batch statistics
(is_training=True everywhere, quirk C1).
"""
import torch

from . import tf_ops as T


def _relu(x):
    return torch.clamp(x, min=0)


def conv_bn(x, W, name, filters, stride, relu=True, rate=1):
    """Network.conv_bn (network.py:172-215): conv(no bias, SAME) -> BN(no affine) -> relu."""
    kind = 'conv2d' if x.dim() == 4 else 'conv3d'
    k = W['%s/%s/kernel' % (name, kind)]
    assert k.shape[-1] == filters
    y = T.batch_norm_train(T.conv(x, k, stride, 'SAME', rate))
    return _relu(y) if relu else y


def deconv_bn(x, W, name, filters, stride=2, relu=True):
    """Network.deconv_bn (network.py:510-550)."""
    k = W['%s/conv3d_transpose/kernel' % name]
    assert k.shape[-2] == filters
    y = T.batch_norm_train(T.conv3d_transpose_same(x, k, stride))
    return _relu(y) if relu else y


def conv(x, W, name, filters, stride=1, relu=True, rate=1):
    """Network.conv (network.py:141-169), biased=False on every call site of the path."""
    k = W['%s/kernel' % name]
    assert k.shape[-1] == filters
    y = T.conv(x, k, stride, 'SAME', rate)
    return _relu(y) if relu else y


def bottleneck(x, W, scope, kernel_size, depth, stride=1, rate=1):
    """Network.bottleneck (network.py:552-602): pre-activation residual unit."""
    depth_in = x.shape[-1]
    preact = _relu(T.batch_norm_train(x, beta=W['%s/preact/beta' % scope]))
    if depth == depth_in:
        if stride == 1:
            shortcut = x
        else:   # slim.max_pool2d([1,1], stride): plain subsampling (never hit on the path)
            shortcut = x[:, ::stride, ::stride]
    else:
        shortcut = T.conv(preact, W['%s/shortcut/weights' % scope], stride, 'SAME',
                          bias=W['%s/shortcut/biases' % scope])
    r = _relu(T.conv(preact, W['%s/conv1/weights' % scope], 1, 'SAME', bias=W['%s/conv1/biases' % scope]))
    if stride == 1:
        r = _relu(T.conv(r, W['%s/conv2/weights' % scope], 1, 'SAME', rate, bias=W['%s/conv2/biases' % scope]))
    else:
        k_eff = kernel_size + (kernel_size - 1) * (rate - 1)
        pb = (k_eff - 1) // 2
        pe = (k_eff - 1) - pb
        r = _relu(T.conv(r, W['%s/conv2/weights' % scope], stride, 'VALID', rate,
                         bias=W['%s/conv2/biases' % scope], explicit_pad=[(pb, pe), (pb, pe)]))
    r = T.conv(r, W['%s/conv3/weights' % scope], 1, 'SAME', bias=W['%s/conv3/biases' % scope])
    return shortcut + r


def res_block(x, W, name, kernel_size, depth, num_block=1, stride=1, rate=1):
    """Network.res_block (network.py:604-616): scopes name_0, name_1, ..., name (last)."""
    if num_block == 1:
        return bottleneck(x, W, name, kernel_size, depth, stride, rate)
    y = bottleneck(x, W, name + '_0', kernel_size, depth, stride, rate)
    for i in range(1, num_block):
        scope = name + '_' + str(i) if i != num_block - 1 else name
        y = bottleneck(y, W, scope, kernel_size, depth, 1, rate)
    return y


def resnet_ds2_spp(x, W, layers=None):
    """ResNetDS2SPP (cnn_wrapper/atvsnet.py:254-292): (B,H,W,3) -> (B,H/4,W/4,32)."""
    bf = 32
    L = {} if layers is None else layers
    y = conv_bn(x, W, 'conv0_0', bf, 2)
    y = conv_bn(y, W, 'conv0_1', bf, 1)
    y = conv_bn(y, W, 'conv0_2', bf, 1)
    y = res_block(y, W, 'conv0_x', 3, bf, 3, 1, 1)
    c1 = res_block(y, W, 'conv1_x', 3, bf * 2, 8, 2, 1)
    y = res_block(c1, W, 'conv2_x', 3, bf * 4, 3, 1, 2)
    c3 = res_block(y, W, 'conv3_x', 3, bf * 4, 3, 1, 4)
    L['conv1_x'], L['conv3_x'] = c1, c3
    h, w = c3.shape[1], c3.shape[2]
    branches = []
    for i, pool in enumerate((64, 32, 16, 8)):
        p = T.avg_pool2d_same(c3, pool, pool)
        p = conv_bn(p, W, 'branch_%d_conv' % i, bf, 1)
        b = T.resize_bilinear_align_corners(p, (h, w))
        L['branch_%d' % i] = b
        branches.append(b)
    cat = torch.cat([c1, c3] + branches, dim=-1)
    y = conv_bn(cat, W, 'fusion0', bf * 4, 1)
    L['fusion0'] = y
    y = conv(y, W, 'fusion1', bf, 1, relu=False)
    L['fusion1'] = y
    return y


def resnet_ds2_spp_shallow_f16(x, W):
    """ResNetDS2SPP_shallow_f16 (cnn_wrapper/atvsnet.py:245-251): (B,H,W,3) -> (B,H/4,W/4,16)."""
    y = res_block(x, W, 'global_refine_conv0_x', 3, 16, 3, 4, 1)
    return conv(y, W, 'global_refine_shallow_feature', 16, 1, relu=False)


def stacked_unet_prob(data, W, layers=None):
    """StackedUNet_prob (cnn_wrapper/atvsnet.py:100-192).

    returns (conv_b2_6_2 (B,D,h,w,1), conv_b2_6_1 (B,D,h,w,8)).
    """
    bf = 8
    L = {} if layers is None else layers
    L['data'] = data
    for b in range(3):
        p = 'conv_b%d_' % b
        if b == 0:
            inp = data
        else:
            q = 'conv_b%d_' % (b - 1)
            inp = L[q + '6_0'] + L[q + '0_1']
            L[p + '0_0'] = inp
        L[p + '1_0'] = conv_bn(inp, W, p + '1_0', bf * 2, 2)
        L[p + '2_0'] = conv_bn(L[p + '1_0'], W, p + '2_0', bf * 4, 2)
        L[p + '3_0'] = conv_bn(L[p + '2_0'], W, p + '3_0', bf * 8, 2)
        L[p + '0_1'] = conv_bn(inp, W, p + '0_1', bf, 1)
        if b == 0:
            i11, i21 = L[p + '1_0'], L[p + '2_0']
        else:
            i11 = L[p + '1_0'] + L[q + '5_0']
            i21 = L[p + '2_0'] + L[q + '4_0']
        L[p + '1_1'] = conv_bn(i11, W, p + '1_1', bf * 2, 1)
        L[p + '2_1'] = conv_bn(i21, W, p + '2_1', bf * 4, 1)
        L[p + '3_1'] = conv_bn(L[p + '3_0'], W, p + '3_1', bf * 8, 1)
        L[p + '4_0'] = deconv_bn(L[p + '3_1'], W, p + '4_0', bf * 4)
        if b == 0:
            i50 = L[p + '4_0'] + L[p + '2_1']
        else:
            i50 = L[p + '4_0'] + L[p + '2_1'] + L['conv_b0_2_1']
        L[p + '5_0'] = deconv_bn(i50, W, p + '5_0', bf * 2)
        if b == 0:
            i60 = L[p + '5_0'] + L[p + '1_1']
        else:
            i60 = L[p + '5_0'] + L[p + '1_1'] + L['conv_b0_1_1']
        L[p + '6_0'] = deconv_bn(i60, W, p + '6_0', bf)
    L['conv_b2_6_1'] = L['conv_b2_6_0'] + L['conv_b2_0_1']
    L['conv_b2_6_2'] = conv(L['conv_b2_6_1'], W, 'conv_b2_6_2', 1, 1, relu=False)
    return L['conv_b2_6_2'], L['conv_b2_6_1']


def cost_vol_refine_net(photo_group, geo_group, prob_vol, vis_hull, W, layers=None):
    """CostVolRefineNet (cnn_wrapper/atvsnet.py:295-336).

    returns (global_refined_cost_vol (B,D,h,w,1), global_refine_3dconv6_1 (B,D,h,w,8)).
    """
    bf = 8
    g = 'global_refine_'
    L = {} if layers is None else layers
    a = conv_bn(photo_group, W, g + 'photo_3dconv', bf, 1)
    b = conv_bn(geo_group, W, g + 'geo_3dconv', bf, 1)
    c = conv_bn(prob_vol, W, g + 'prob_3dconv', bf, 1)
    d = conv_bn(vis_hull, W, g + 'vishull_3dconv', bf, 1)
    cat = torch.cat([a, b, c, d], dim=-1)
    L[g + 'concat'] = cat
    c10 = conv_bn(cat, W, g + '3dconv1_0', bf * 2, 2)
    c20 = conv_bn(c10, W, g + '3dconv2_0', bf * 4, 2)
    c30 = conv_bn(c20, W, g + '3dconv3_0', bf * 8, 2)
    c01 = conv_bn(cat, W, g + '3dconv0_1', bf, 1)
    c11 = conv_bn(c10, W, g + '3dconv1_1', bf * 2, 1)
    c21 = conv_bn(c20, W, g + '3dconv2_1', bf * 4, 1)
    c31 = conv_bn(c30, W, g + '3dconv3_1', bf * 8, 1)
    c40 = deconv_bn(c31, W, g + '3dconv4_0', bf * 4)
    c50 = deconv_bn(c40 + c21, W, g + '3dconv5_0', bf * 2)
    c60 = deconv_bn(c50 + c11, W, g + '3dconv6_0', bf)
    c61 = c60 + c01
    L[g + '3dconv6_1'] = c61
    out = conv(c61, W, 'global_refined_cost_vol', 1, 1, relu=False)
    return out, c61


def attention_aggregation(X, W, name):
    """Network.attention_aggregation with second_weight=True, relu=True, biased=False
    (network.py:282-351,378-408; call sites atvsnet.py:202,234).

    X: (B,D,h,w,C,Nv) -> (B,D,h,w,C).
    """
    wu = W['%s/attention_activation/weight_unique' % name]
    ws = W['%s/attention_activation/weight_shared' % name]
    nv = X.shape[-1]
    S = [_relu(T.conv(X[..., n], ws, 1, 'SAME')) for n in range(nv)]
    S_sum = S[0]
    for n in range(1, nv):
        S_sum = S_sum + S[n]
    U = [(_relu(T.conv(X[..., n], wu, 1, 'SAME')) - S[n]) + S_sum for n in range(nv)]
    score = T.softmax(torch.stack(U, dim=-1), axis=-1)
    return (score * X).sum(dim=-1)


def output_conv(cost_volume, W, name='attention_prob_vol'):
    """OutputConv / OutputConv_refine (atvsnet.py:216-226) + squeeze (model.py:132-140)."""
    return conv(cost_volume, W, name, 1, 1, relu=False).squeeze(-1)
