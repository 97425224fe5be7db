"""viettts_amd — MI355X-native mel->waveform hot path of NTT123/vietTTS.

Only the path BASELINE.json's north_star names lives here: the HiFi-GAN V1
generator (reference: vietTTS/hifigan/model.py:77-125) behind the reference's
call surface ``mel2wave(mel)`` (vietTTS/hifigan/mel2wave.py:20), computed by
hand-written gfx950 HIP kernels reached through a C-ABI shared library
(include/vtts_hifigan.h).  PyTorch-ROCm is used for device memory, streams
and torch.distributed (RCCL) only.
"""

__version__ = "0.1.0"
