"""CPU oracle (test infrastructure only).  See dalle_oracle.py / vae_oracle.py headers."""
