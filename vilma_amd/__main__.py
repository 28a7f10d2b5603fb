from .frontend import main

main()
