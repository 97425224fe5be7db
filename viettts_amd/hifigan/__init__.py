"""HiFi-GAN generator hot path (reference package: vietTTS/hifigan)."""
