"""DiffVC decoder on MI355X: drop-in mirror of DiffVC/model/{base,modules,diffusion}.py (decoder only; the mel
encoder / post-net / speaker encoder of DiffVC are out of the accelerated scope)."""
