"""Drop-in import surface of NTT123/vietTTS for the paths this repository implements on MI355X.

``from vietTTS.hifigan.mel2wave import mel2wave``, ``from vietTTS.nat.text2mel import text2mel`` and
``python -m vietTTS.synthesizer --text ...`` resolve to viettts_amd (hand-written HIP behind include/*.h); nothing here
holds logic.  Everything else the reference package contains (trainers, data loaders, DSP, notebooks) is out of scope
and absent on purpose: importing it raises ImportError instead of silently running something else.
"""
