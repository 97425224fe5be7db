"""NAT text->mel front end (reference package: vietTTS/nat).  Call surface only."""
