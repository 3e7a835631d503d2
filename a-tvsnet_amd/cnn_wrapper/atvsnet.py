"""Layer graphs of A-TVSNet on the Network operator API.

Same class names, layer names (hence variable names) and wiring as
/root/reference/cnn_wrapper/atvsnet.py; the three stacked U-Nets are written as a
loop over the stack index instead of being spelled out three times.
Exact graphs: SURVEY.md Appendix A.
"""
from .network import Network


def _stacked_unet(net, with_prob_head):
    """Three 3-level 3-D U-Nets with cross-stack skip adds (reference atvsnet.py:5-96 / :100-192).

    Stack b reads I_b (I_0 = 'data'; I_b = conv_b{b-1}_6_0 + conv_b{b-1}_0_1).  Layer
    conv_b{b}_{level}_{0|1}: level 1-3 stride-2 encoders (_0) and same-resolution convs (_1),
    level 4-6 stride-2 transposed-conv decoders.
    """
    f = 8
    for b in range(3):
        n = lambda s: 'conv_b%d_%s' % (b, s)          # noqa: E731
        p = lambda s: 'conv_b%d_%s' % (b - 1, s)      # noqa: E731
        if b == 0:
            src = 'data'
        else:
            src = n('0_0')
            net.feed(p('6_0'), p('0_1')).add(name=src, defer=True)     # formed on load by the siblings below
        # defer_bn: the layer's batch norm (+ ReLU) is left to its consumers -- `add`s, which normalise on the fly, and the
        # convolutions that normalise while they stage their halo (ops.norm_on_load_3d_ok); anything else materialises it.
        # conv_b*_0_1 and the encoder branch conv_b*_1_0 read the same tensor: issued as siblings (one launch)
        (net.feed(src)
            .conv_bn_siblings(dict(kernel_size=3, filters=f, strides=1, name=n('0_1'), defer_bn=True),
                              dict(kernel_size=3, filters=f * 2, strides=2, name=n('1_0'), defer_bn=True))
            .conv_bn(3, f * 4, 2, name=n('2_0'), defer_bn=True)
            .conv_bn(3, f * 8, 2, name=n('3_0'), defer_bn=True))
        if b == 0:
            net.feed(n('1_0')).conv_bn(3, f * 2, 1, name=n('1_1'), defer_bn=True)
            net.feed(n('2_0')).conv_bn(3, f * 4, 1, name=n('2_1'), defer_bn=True)
        else:
            # (the half-resolution skip sum is formed on load by conv_b*_1_1's kernel)
            net.feed(n('1_0'), p('5_0')).add(name=n('1_1_concat'), defer=True).conv_bn(3, f * 2, 1, name=n('1_1'), defer_bn=True)
            net.feed(n('2_0'), p('4_0')).add(name=n('2_1_concat')).conv_bn(3, f * 4, 1, name=n('2_1'), defer_bn=True)
        (net.feed(n('3_0'))
            .conv_bn(3, f * 8, 1, name=n('3_1'))
            .deconv_bn(3, f * 4, 2, name=n('4_0'), defer_bn=True))
        skip2 = [n('4_0'), n('2_1')] + ([] if b == 0 else ['conv_b0_2_1'])
        net.feed(*skip2).add(name=n('4_1')).deconv_bn(3, f * 2, 2, name=n('5_0'), defer_bn=True)
        skip1 = [n('5_0'), n('1_1')] + ([] if b == 0 else ['conv_b0_1_1'])
        net.feed(*skip1).add(name=n('5_1'), defer=True).deconv_bn(3, f, 2, name=n('6_0'), defer_bn=True)     # summed on load
    net.feed('conv_b2_6_0', 'conv_b2_0_1').add(name='conv_b2_6_1')
    if with_prob_head:
        net.conv(3, 1, 1, relu=False, name='conv_b2_6_2')


class StackedUNet(Network):
    """Cost-volume regulariser without the probability head (reference atvsnet.py:5-96)."""

    def setup(self):
        _stacked_unet(self, with_prob_head=False)


class StackedUNet_prob(Network):
    """Regulariser with the 8->1 head: outputs conv_b2_6_1 (filtered cost) and conv_b2_6_2
    (reference atvsnet.py:100-192)."""

    def setup(self):
        _stacked_unet(self, with_prob_head=True)


def _attention(net, scope, head):
    # 'data': (B,D,H,W,C,N) or a list of N (B,D,H,W,C) tensors, N = number of source views
    shape_in = net.get_shape_by_name('data')
    net.feed('data').attention_aggregation(kernel_size=3, name=scope, second_weight=True, relu=True, biased=False,
                                           n_view=shape_in[-1])
    if head:
        net.conv(3, 1, 1, relu=False, name=head)


class AttAggregation_keepchannel(Network):
    """AAM1 attention (reference atvsnet.py:196-203)."""

    def setup(self):
        _attention(self, 'attention_aggregate', None)


class AttAggregation(Network):
    """AAM1 attention + 8->1 head (reference atvsnet.py:206-213)."""

    def setup(self):
        _attention(self, 'attention_aggregate', 'attention_prob_vol')


class OutputConv(Network):
    """8->1 convolution after AAM1 (reference atvsnet.py:216-220)."""

    def setup(self):
        self.feed('data').conv(3, 1, 1, relu=False, name='attention_prob_vol')


class OutputConv_refine(Network):
    """8->1 convolution after AAM2 (reference atvsnet.py:222-226)."""

    def setup(self):
        self.feed('data').conv(3, 1, 1, relu=False, name='attention_prob_vol_refine')


class AttAggregation_refine_keepchannel(Network):
    """AAM2 attention (reference atvsnet.py:229-234)."""

    def setup(self):
        _attention(self, 'attention_aggregate_refine', None)


class AttAggregation_refine(Network):
    """AAM2 attention + head (reference atvsnet.py:236-242)."""

    def setup(self):
        _attention(self, 'attention_aggregate_refine', 'attention_prob_vol_refine')


class ResNetDS2SPP_shallow_f16(Network):
    """Low-level 16-channel features at 1/4 resolution for the refinement (reference atvsnet.py:245-251)."""

    def setup(self):
        (self.feed('data')
             .res_block(3, 16, num_block=3, stride=4, rate=1, name='global_refine_conv0_x')
             .conv(1, 16, 1, relu=False, name='global_refine_shallow_feature'))


class ResNetDS2SPP(Network):
    """2-D feature tower: dilated ResNet, 1/4 resolution, spatial pyramid pooling, 32 channels
    (reference atvsnet.py:254-292)."""

    def setup(self):
        f = 32
        (self.feed('data')
             .conv_bn(3, f, 2, name='conv0_0', defer_bn=True)      # normalised while conv0_1 / conv0_2 / fusion1 stage their input
             .conv_bn(3, f, 1, name='conv0_1', defer_bn=True)
             .conv_bn(3, f, 1, name='conv0_2')
             .res_block(3, f, num_block=3, stride=1, rate=1, name='conv0_x')
             .res_block(3, f * 2, num_block=8, stride=2, rate=1, name='conv1_x')
             .res_block(3, f * 4, num_block=3, stride=1, rate=2, name='conv2_x')
             .res_block(3, f * 4, num_block=3, stride=1, rate=4, name='conv3_x'))
        size = self.get_shape_by_name('conv3_x')[1:3]
        # (the four pyramid branches as parallel graph branches on side streams: measured SLOWER, 20.02 -> 20.3-20.7 ms per map --
        # the fork / join of a captured graph costs more than the launch-bound chains on tiny pooled maps gain)
        self.concat_buffer('concat_feature', self.layers['conv3_x'], 10 * f)     # [conv1_x 2f | conv3_x 4f | 4 branches of f]
        for i, pool in enumerate((64, 32, 16, 8)):
            (self.feed('conv3_x')
                 .avg_pool(pool, pool, name='branch_%d_pool' % i)
                 .conv_bn(3, f, 1, relu=True, name='branch_%d_conv' % i)
                 .image_resize(size=size, method='bilinear', name='branch_%d' % i, align_corners=True,
                               out_slice=('concat_feature', (6 + i) * f)))      # straight into its slice of the concat
        (self.feed('conv1_x', 'conv3_x', 'branch_0', 'branch_1', 'branch_2', 'branch_3')
             .concat(axis=-1, name='concat_feature')
             .conv_bn(3, f * 4, 1, relu=True, name='fusion0', defer_bn=True)
             .conv(1, f, 1, relu=False, name='fusion1'))


class CostVolRefineNet(Network):
    """Refinement network over the photo / geo / probability / visual-hull volumes
    (reference atvsnet.py:295-336)."""

    def setup(self):
        f = 8
        g = 'global_refine_'
        # the four stems + their concat: one HBM-bound pass over the 32-channel buffer where the inputs allow it
        # (Network.refine_stems), else conv_bn per stem into its slice of the buffer
        (self.refine_stems(g + 'concat', [('photo_group', g + 'photo_3dconv'), ('geo_group', g + 'geo_3dconv'),
                                          ('prob_vol', g + 'prob_3dconv'), ('vis_hull', g + 'vishull_3dconv')], f)
             .conv_bn_siblings(dict(kernel_size=3, filters=f, strides=1, name=g + '3dconv0_1', defer_bn=True),
                               dict(kernel_size=3, filters=f * 2, strides=2, name=g + '3dconv1_0', defer_bn=True))
             .conv_bn(3, f * 4, 2, name=g + '3dconv2_0', defer_bn=True)
             .conv_bn(3, f * 8, 2, name=g + '3dconv3_0', defer_bn=True))
        self.feed(g + '3dconv1_0').conv_bn(3, f * 2, 1, name=g + '3dconv1_1', defer_bn=True)
        self.feed(g + '3dconv2_0').conv_bn(3, f * 4, 1, name=g + '3dconv2_1', defer_bn=True)
        (self.feed(g + '3dconv3_0')
             .conv_bn(3, f * 8, 1, name=g + '3dconv3_1')
             .deconv_bn(3, f * 4, 2, name=g + '3dconv4_0', defer_bn=True))
        (self.feed(g + '3dconv4_0', g + '3dconv2_1')
             .add(name=g + '3dconv4_1')
             .deconv_bn(3, f * 2, 2, name=g + '3dconv5_0', defer_bn=True))
        (self.feed(g + '3dconv5_0', g + '3dconv1_1')
             .add(name=g + '3dconv5_1', defer=True)          # summed on load by the transposed convolution
             .deconv_bn(3, f, 2, name=g + '3dconv6_0', defer_bn=True))
        # extension: with an input 'residual_base' (the aggregated cost volume every source's residual is added to, model.py:438)
        # the layer global_refine_3dconv6_1_plus = residual_base + global_refine_3dconv6_1 comes out of the same pass
        (self.feed(g + '3dconv6_0', g + '3dconv0_1')
             .add(name=g + '3dconv6_1', plus=('residual_base' if 'residual_base' in self.layers else None))
             .conv(3, 1, 1, relu=False, name='global_refined_cost_vol'))
