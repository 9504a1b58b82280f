"""The two helpers of Grad-TTS/hifi-gan/xutils.py the generator needs (:23-36)."""


def init_weights(m, mean=0.0, std=0.01):
    """xutils.py:23-26 -- normal(mean, std) init of every *Conv* module's weight."""
    if "Conv" in m.__class__.__name__:
        m.weight.data.normal_(mean, std)


def get_padding(kernel_size, dilation=1):
    """xutils.py:35-36 -- 'same' padding of a dilated odd kernel."""
    return (kernel_size * dilation - dilation) // 2
