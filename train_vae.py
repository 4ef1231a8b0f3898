"""train_vae.py -- the reference's train_vae.py imports a non-existent symbol (train_vae.py:8); the working
upstream entry is train_vae_tf.py.  Same program, same CLI:
    python train_vae.py --model vae_example [--new]"""
from train_vae_tf import main, parse_args  # noqa: F401

if __name__ == "__main__":
    main()
