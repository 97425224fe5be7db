"""Constants of the NAT front end that the call surface needs (reference: vietTTS/nat/config.py:8-59).

What ``text2tokens``, the frame-count arithmetic, the duration model, the acoustic model's inference path
(``viettts_amd/nat/acoustic.py``) and the CLI read is mirrored here; training knobs are out of scope.
"""
from pathlib import Path


class FLAGS:
    # model dimensions (config.py:11-17)
    duration_lstm_dim = 256
    vocab_size = 256
    postnet_dim = 512
    acoustic_decoder_dim = 512
    acoustic_encoder_dim = 256
    # Montreal-Forced-Aligner specials: [sil] [sp] [spn] [word end]   (config.py:24-27)
    special_phonemes = ["sil", "sp", "spn", " "]
    sil_index = 0
    sp_index = 0
    word_end_index = 3
    # letter-level phoneme inventory, order defines the token ids (config.py:28-39)
    _normal_phonemes = list(
        "abcdeghikl" "mnopqrstuv" "xyàáâãèéêì" "íòóôõùúýăđ" "ĩũơưạảấầẩẫ"
        "ậắằẳẵặẹẻẽế" "ềểễệỉịọỏốồ" "ổỗộớờởỡợụủ" "ứừửữựỳỵỷỹ"
    )
    # dsp (config.py:42-47): 16 kHz audio, hop = n_fft // 4 = 256 samples per mel frame
    mel_dim = 80
    n_fft = 1024
    sample_rate = 16000
    # checkpoints / data (config.py:57-58)
    ckpt_dir = Path("assets/infore/nat")
    data_dir = Path("train_data")


def load_phonemes_set():
    """special + normal phonemes (vietTTS/nat/data_loader.py:11-13)."""
    return FLAGS.special_phonemes + FLAGS._normal_phonemes
